"""one denoiser pass (steps = 1): B = 1 vs batch-of-8 pair 0 -- projected features, conf"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import T, weights, pair
DEV = "cuda:0"
variant, N, M, steps, mc = "4dmatch", 512, 512, 1, 40
v = synth.VARIANTS[variant]
cases = [(470, 391, 62), (512, 512, 61), (500, 480, 63), (512, 300, 64), (333, 512, 65), (450, 450, 66), (512, 511, 69), (400, 390, 68)]
prs = [pair(variant, N, M, c[2])[1] for c in cases]
ms = torch.stack([torch.arange(N) < c[0] for c in cases]); mt = torch.stack([torch.arange(M) < c[1] for c in cases])
noise = torch.stack([T(synth.step_noise(N, M, c[2], steps)) for c in cases], 1)
def run(sel, planes=True):
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                        sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, planes=planes)
    cat = lambda k: torch.cat([prs[i][k] for i in sel]).to(DEV)
    o = eng.run(cat("f_s"), cat("f_t"), cat("p_s"), cat("p_t"), cat("x_T"), ms[sel].to(DEV), mt[sel].to(DEV), noise=noise[:, sel].to(DEV), trace=True, side_outputs=True)
    return {k: o[k].cpu().clone() for k in o if torch.is_tensor(o[k])}
a = run([0]); b = run(list(range(8))); c = run([0], planes=False)
for nm, x, y in (("B1 vs B8", a, b), ("B1 planes vs B1 f32", a, c)):
    print(nm)
    for k in ("src_feats", "tgt_feats", "src_feats_nopos", "tgt_feats_nopos", "conf_matrix_pred", "x0"):
        xa = x[k][0].double() if k != "x0" else x[k][0, 0].double(); ya = y[k][0].double() if k != "x0" else y[k][0, 0].double()
        d = (xa - ya).abs()
        print("  ", k, "max diff %.3e" % d.max(), "max val %.3f" % xa.abs().max(), "nonzero diffs", int((d > 0).sum()), "of", d.numel())
