"""bf16x3-split GEMM vs the f32-MFMA GEMM: error against an fp64 product, and time per launch (graph-timed)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = "cuda:0"
CFGS = [int(c) for c in os.environ.get("CFGS", "9,50,60,61").split(",")]
shapes = [(8192, 432, 432), (8192, 432, 864), (16384, 432, 432), (8192, 864, 864), (1000, 432, 436), (256, 432, 432)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
torch.manual_seed(0)
for rows, ncols, K in shapes:
    x = torch.randn(rows, K, device=dev) * 3; W = torch.randn(ncols, K, device=dev) / K ** 0.5
    ref = x.double() @ W.double().T
    Wp = lib.pack_weight(W)
    def run(cfg):
        return lib.linear_packed(x, W, Wp) if cfg >= 50 else lib.linear(x, W)
    print("%6d x %4d x %4d" % (rows, ncols, K))
    for cfg in CFGS:
        lib.raw().dr_debug_gemm_config(cfg)
        y = run(cfg)
        torch.cuda.synchronize()
        err = (y.double() - ref).abs()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50): run(cfg)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print("   cfg%2d %7.1f us %6.1f TF   max err %.2e mean err %.2e" % (cfg, us, 2.0 * rows * ncols * K / us / 1e6, err.max().item(), err.mean().item()))
lib.raw().dr_debug_gemm_config(-1)
