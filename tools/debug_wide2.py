import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
torch.manual_seed(0)
for rows in (256, 8192):
    for ncols, K in ((528, 528), (1056, 1056), (528, 1056), (432, 432), (864, 864)):
        x = torch.randn(rows, K, device="cuda"); W = torch.randn(ncols, K, device="cuda") / K ** 0.5
        Wp = lib.pack_weight(W)
        lib.raw().dr_debug_gemm_wide_min(1)
        y = lib.linear_packed(x, W, Wp)
        lib.raw().dr_debug_gemm_wide_min(-1)
        ref = x.double() @ W.double().t()
        err = (y.double() - ref).abs()
        print(rows, ncols, K, "max err %.2e" % err.max().item(), "bad cols", torch.nonzero(err.amax(0) > 1e-3).flatten()[:8].tolist(), "bad rows", torch.nonzero(err.amax(1) > 1e-3).flatten()[:8].tolist())
