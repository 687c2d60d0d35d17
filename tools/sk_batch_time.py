"""Time the Sinkhorn of a batch of large tiles (cfg5's 8 x 1024 x 2048 per call): the batch form of the co-resident kernel (whole batch in
registers, one launch) against the multi-launch grid form (DR_SK_BATCH=0), HIP events, the three type pairs of the loop."""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
lib.raw().dr_debug_enable_env(1)
res = []
for batch in (1, 0):
    os.environ["DR_SK_BATCH"] = str(batch)
    for (B, N, M, dt, o32) in ((8, 1024, 2048, torch.float64, True), (8, 1024, 2048, torch.float32, False), (8, 1024, 2048, torch.float64, False),
                               (5, 1000, 1530, torch.float32, False), (4, 1024, 2048, torch.float32, False)):
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
        us = e0.elapsed_time(e1) / 20 * 1e3
        byts = B * N * M * (x.element_size() + out.element_size())
        res.append(dict(batch_form=batch, B=B, N=N, M=M, dtype_in=str(dt), dtype_out=str(out.dtype), us_per_call=us, algorithmic_GBps=byts / us / 1e3,
                        frac_of_8TBps=byts / us / 1e3 / 8000.0))
        print(res[-1], flush=True)
print(json.dumps(res))
