"""Time the multi-launch (grid) Sinkhorn on batches of large tiles (the batched 2D-3D loop: 8 x 1024 x 2048), vector vs scalar accesses."""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
res = []
for vec in (1, 0):
    lib.raw().dr_debug_enable_env(1)
    os.environ["DR_SK_GRID_VEC"] = str(vec)
    for (B, N, M, dt, o32) in ((8, 1024, 2048, torch.float64, True), (8, 1024, 2048, torch.float32, False), (8, 1024, 2048, torch.float64, False), (16, 1024, 2048, torch.float32, False),
                               (32, 512, 512, torch.float32, False)):
        x = (torch.randn(B, N, M, device="cuda") * 2).to(dt)
        a = torch.tensor(1.0, device="cuda")
        out = lib.sinkhorn(x, a, 3, out_f32=o32)
        for _ in range(3):
            lib.sinkhorn(x, a, 3, out_f32=o32, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.sinkhorn(x, a, 3, out_f32=o32, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        byts = B * N * M * (x.element_size() + out.element_size())
        res.append(dict(vec=vec, B=B, N=N, M=M, dtype_in=str(dt), dtype_out=str(out.dtype), us_per_call=us, algorithmic_GBps=byts / us / 1e3))
        print(res[-1], flush=True)
print(json.dumps(res))
