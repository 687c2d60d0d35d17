import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
rows, ncols, K, cfg = [int(a) for a in sys.argv[1:5]]
x = torch.randn(rows, K, device="cuda:0"); W = torch.randn(ncols, K, device="cuda:0") / K ** 0.5
lib.raw().dr_debug_gemm_config(cfg)
for _ in range(5): lib.linear(x, W)
torch.cuda.synchronize()
