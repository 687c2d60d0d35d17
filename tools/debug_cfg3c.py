"""which kernel family breaks the 8-pair 512^2 4D denoiser evaluation: batched vs single, with kernel overrides"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth, lib
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import T, weights, pair
DEV = "cuda:0"
variant, N, M = "4dmatch", 512, 512
v = synth.VARIANTS[variant]
eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=1,
                    sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=40, n_layers=v["n_layers"], device=DEV)
cases = [(470, 391, 62), (512, 512, 61), (500, 480, 63), (512, 300, 64), (333, 512, 65), (450, 450, 66), (512, 511, 67), (400, 390, 68)]
prs = [pair(variant, N, M, c[2])[1] for c in cases]
cat = lambda k: torch.cat([q[k] for q in prs]).to(DEV)
ms = torch.stack([torch.arange(N) < c[0] for c in cases]).to(DEV); mt = torch.stack([torch.arange(M) < c[1] for c in cases]).to(DEV)
singles = []
for i, q in enumerate(prs):
    so, to, conf = eng.denoise_match(q["f_s"].to(DEV), q["f_t"].to(DEV), q["p_s"].to(DEV), q["p_t"].to(DEV), ms[i:i + 1], mt[i:i + 1])
    singles.append((so.clone(), to.clone(), conf.clone()))
def batch(tag):
    so, to, conf = eng.denoise_match(cat("f_s"), cat("f_t"), cat("p_s"), cat("p_t"), ms, mt)
    ds = max((so[i] - singles[i][0][0]).abs().max().item() for i in range(8))
    dt = max((to[i] - singles[i][1][0]).abs().max().item() for i in range(8))
    dc = max((conf[i] - singles[i][2][0]).abs().max().item() for i in range(8))
    print(tag, "max |src feats| dev %.2e  tgt %.2e  conf %.2e" % (ds, dt, dc))
batch("default        ")
lib.raw().dr_debug_gemm_wide_min(1000000); batch("no packed gemm ")
lib.raw().dr_debug_gemm_wide_min(-1); lib.raw().dr_debug_attention_config(1000000); batch("no flash attn  ")
lib.raw().dr_debug_gemm_wide_min(1000000); batch("neither        ")
