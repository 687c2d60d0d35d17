"""batch-form Sinkhorn (8 x 1024 x 2048) with a tile's workgroups on one XCD (DR_SK_XCD=1, default) against spread over the 8 XCDs (=0);
interleaved rounds in one process; needs the debug env gate (enabled here)"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
lib.raw().dr_debug_enable_env(1)
res = {}
cases = ((8, 1024, 2048, torch.float64, True), (8, 1024, 2048, torch.float32, False), (16, 512, 2048, torch.float32, False), (4, 1024, 2048, torch.float32, False))
outs = {}
for rnd in range(3):
    for xcd in ("1", "0"):
        os.environ["DR_SK_XCD"] = xcd
        for (B, N, M, dt, o32) in cases:
            torch.manual_seed(1)
            x = (torch.randn(B, N, M, device="cuda") * 2).to(dt)
            a = torch.tensor(1.0, device="cuda")
            out = lib.sinkhorn(x, a, 3, out_f32=o32)
            for _ in range(5):
                lib.sinkhorn(x, a, 3, out_f32=o32, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.sinkhorn(x, a, 3, out_f32=o32, out=out)
            e1.record()
            torch.cuda.synchronize()
            lib.device_status("cuda:0")
            key = "%dx%dx%d %s xcd_local=%s" % (B, N, M, str(dt).split(".")[-1], xcd)
            res.setdefault(key, []).append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
            k2 = (B, N, M, str(dt))
            if k2 in outs: assert torch.equal(outs[k2], out), "placement changed the bits"
            else: outs[k2] = out.clone()
for k, v in res.items(): print(k, v)
print(json.dumps(res))
