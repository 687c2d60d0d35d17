"""Small (single-pair) GEMM shapes: the latency kernel (auto) vs the tiled f32-MFMA kernels; graph-timed."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = "cuda:0"
SHAPES = [(256, 432, 432), (512, 432, 432), (256, 864, 864), (512, 864, 864), (256, 432, 864), (512, 432, 864), (1024, 432, 432), (2048, 432, 432), (256, 256, 432)]
if os.environ.get("SHAPES") == "mid":     # the 2D-3D loop (3 072 token rows, C = 256), a real-size pair (1 193 rows, C = 432), 2-4 pairs of 256 x 256
    SHAPES = [(3072, 256, 256), (3072, 512, 256), (3072, 256, 512), (2048, 256, 256), (1024, 256, 256), (3072, 768, 256),
              (1193, 432, 432), (1193, 864, 864), (1193, 432, 864), (1193, 1296, 432), (1024, 432, 432), (1024, 864, 864), (2048, 864, 864), (2048, 432, 864)]
for rows, ncols, K in SHAPES:
    x = torch.randn(rows, K, device=dev); W = torch.randn(ncols, K, device=dev) / K ** 0.5
    ref = x.double() @ W.double().T
    line = "%5d x %4d x %4d :" % (rows, ncols, K)
    for cfg in ((-1, 0, 1, 2, 9) if os.environ.get("SHAPES") == "mid" else (-1, 0, 9, 11, 12)):
        lib.raw().dr_debug_gemm_config(cfg)
        y = lib.linear(x, W); torch.cuda.synchronize()
        err = (y.double() - ref).abs().max().item()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50): lib.linear(x, W)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        line += "  cfg%2d %5.1f us (err %.1e)" % (cfg, e0.elapsed_time(e1) / 200 * 1e3, err)
    print(line)
lib.raw().dr_debug_gemm_config(-1)
