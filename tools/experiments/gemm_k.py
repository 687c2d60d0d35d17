import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = "cuda:0"
for rows, ncols, K in [(8192, 896, 864), (8192, 896, 8640), (16384, 1792, 8640), (4096, 896, 8640)]:
    x = torch.randn(rows, K, device=dev); W = torch.randn(ncols, K, device=dev) / K ** 0.5
    line = "%6d x %4d x %5d :" % (rows, ncols, K)
    for cfg in (2, 4, 7, 8):
        lib.raw().dr_debug_gemm_config(cfg)
        for _ in range(2): lib.linear(x, W)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lib.linear(x, W)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        line += "  cfg%2d %8.1f us %6.1f TF" % (cfg, us, 2.0 * rows * ncols * K / us / 1e6)
    print(line)
