"""dr_sinkhorn_f16 against dr_sinkhorn_f32 at the headline's size (B tiles of 256 x 256, 3 iterations): us per call and algorithmic GB/s (HIP events)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
a = torch.tensor(1.0, device=dev)
for B in (512, 4096):
    x = torch.randn(B, 256, 256, device=dev) * 3
    xh = x.half()
    def t(f, n=20):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t32, t16 = t(lambda: lib.sinkhorn(x, a, 3)), t(lambda: lib.sinkhorn_f16(xh, a, 3))
    print("B %4d  f32 %7.1f us (%4.2f TB/s)   f16 %7.1f us (%4.2f TB/s of its own bytes)   x %.2f" % (B, t32, B * 65536 * 8 / t32 / 1e6, t16, B * 65536 * 4 / t16 / 1e6, t32 / t16))
