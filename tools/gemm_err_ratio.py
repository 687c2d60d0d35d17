import sys, os, torch
sys.path.insert(0, "/root/repo/diff-reg_amd"); sys.path.insert(0, "/root/repo")
from diffreg_hip import lib, synth
from tests.helpers import T
DEV="cuda"
for rows, ncols, K in [(1000, 432, 432), (4096, 432, 864), (777, 864, 864), (130, 224, 16), (8192, 432, 432)]:
    x = (T(synth.hash_normal(5, rows + K, (rows, K))).float() * 3).to(DEV)
    W = (T(synth.hash_uniform(6, ncols + K, (ncols, K))).float() / K ** 0.5).to(DEV)
    Wp = lib.pack_weight(W); ref = x.double() @ W.double().T
    out = []
    for cfg in (-1, 0, 9, 50, 61):
        lib.raw().dr_debug_gemm_config(cfg)
        y = lib.linear_packed(x, W, Wp) if cfg >= 50 else lib.linear(x, W)
        e = (y.double() - ref).abs()
        out.append("cfg%d mean %.2e max %.2e" % (cfg, e.mean().item(), e.max().item()))
    # torch fp32 matmul on GPU for reference
    e = ((x @ W.T).double() - ref).abs(); out.append("torch mean %.2e max %.2e" % (e.mean().item(), e.max().item()))
    print(rows, ncols, K, " | ".join(out), "scale %.1f" % ref.abs().max().item())
lib.raw().dr_debug_gemm_config(-1)
