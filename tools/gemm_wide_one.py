"""One shape of the packed (wide split) GEMM, a few launches: the target of rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
rows, ncols, K = [int(v) for v in os.environ.get("SHAPE", "32768x432x432").split("x")]
x = torch.randn(rows, K, device="cuda"); W = torch.randn(ncols, K, device="cuda") / K ** 0.5
Wp = lib.pack_weight(W)
lib.raw().dr_debug_gemm_config(int(os.environ.get("CFG", "50")))
for _ in range(int(os.environ.get("N", "10"))):
    y = lib.linear_packed(x, W, Wp)
torch.cuda.synchronize()
