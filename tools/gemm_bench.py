"""dr_linear_f32 on the loop's GEMM shapes, every tile configuration: us and TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = "cuda:0"
CFGS = [int(c) for c in os.environ.get("CFGS", "0,1,2,9,-1").split(",")]
shapes = [(16384, 432, 432), (16384, 432, 864), (4096, 432, 432), (4096, 864, 864), (4096, 432, 864), (256, 432, 432), (512, 432, 432), (512, 864, 864), (512, 432, 864), (256, 864, 864), (256, 256, 432),
          (2048, 432, 432), (2048, 864, 864), (8192, 432, 432), (8192, 864, 864), (8192, 432, 864), (16384, 864, 864)]
for rows, ncols, K in shapes:
    x = torch.randn(rows, K, device=dev); W = torch.randn(ncols, K, device=dev) / K ** 0.5
    line = "%6d x %4d x %4d :" % (rows, ncols, K)
    for cfg in CFGS:
        lib.raw().dr_debug_gemm_config(cfg)
        for _ in range(3): lib.linear(x, W)
        torch.cuda.synchronize()
        # 50 launches inside one HIP graph: GPU-side time per launch incl. the ~1.5 us kernel boundary
        g = torch.cuda.CUDAGraph()
        s_ = torch.cuda.Stream()
        with torch.cuda.stream(s_):
            lib.linear(x, W)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(50): lib.linear(x, W)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        line += "  cfg%2d %7.1f us %6.1f TF" % (cfg, us, 2.0 * rows * ncols * K / us / 1e6)
    print(line)
lib.raw().dr_debug_gemm_config(-1)
