"""dr_linear_packed_f32 vs dr_linear_f32 (f32-MFMA kernel) with every epilogue, ragged shapes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
torch.manual_seed(1)
dev = "cuda"
for rows, ncols, K in [(1000, 432, 432), (8192, 432, 432), (777, 864, 864), (130, 224, 16), (129, 228, 440), (5, 4, 8), (4096, 432, 864)]:
    x = torch.randn(rows, K, device=dev); W = torch.randn(ncols, K, device=dev) / K ** 0.5
    Wp = lib.pack_weight(W)
    ref64 = x.double() @ W.double().T
    for epi, name in [(0, "none"), (1, "relu"), (2, "rotary"), (3, "relu+rotary")]:
        kw = {}
        if epi & 2:
            if ncols % 4: continue
            ang = torch.rand(rows, ncols // 2, device=dev) * 6.28
            kw = dict(cos=ang.cos().contiguous(), sin=ang.sin().contiguous(), rot_C=ncols)
        lib.raw().dr_debug_gemm_config(-1)
        a = lib.linear(x, W, epilogue=epi, scale=0.37, **kw)
        lib.raw().dr_debug_gemm_config(int(os.environ.get("CFG", "50")))
        b = lib.linear_packed(x, W, Wp, epilogue=epi, scale=0.37, **kw)
        lib.raw().dr_debug_gemm_config(-1)
        d = (a - b).abs().max().item()
        print("%5d x %4d x %4d %-12s max|f32mfma - split| = %.2e  (|out| max %.1f)" % (rows, ncols, K, name, d, a.abs().max().item()))
        assert d < 5e-5, d
print("ok")
