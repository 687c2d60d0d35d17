"""B = 1 vs batch-of-8 (pair 0), plane path: per-step deviation of x0 / R / t / cond between the two runs"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import T, weights, pair
DEV = "cuda:0"
variant, N, M, steps, mc = "4dmatch", 512, 512, 20, 40
v = synth.VARIANTS[variant]
cases = [(470, 391, 62), (512, 512, 61), (500, 480, 63), (512, 300, 64), (333, 512, 65), (450, 450, 66), (512, 511, 69), (400, 390, 68)]
prs = [pair(variant, N, M, c[2])[1] for c in cases]
ms = torch.stack([torch.arange(N) < c[0] for c in cases]); mt = torch.stack([torch.arange(M) < c[1] for c in cases])
noise = torch.stack([T(synth.step_noise(N, M, c[2], steps)) for c in cases], 1)
def run(sel):
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                        sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, planes=True)
    cat = lambda k: torch.cat([prs[i][k] for i in sel]).to(DEV)
    o = eng.run(cat("f_s"), cat("f_t"), cat("p_s"), cat("p_t"), cat("x_T"), ms[sel].to(DEV), mt[sel].to(DEV), noise=noise[:, sel].to(DEV), trace=True)
    return {k: o[k].cpu().clone() for k in ("x0", "R_forwd", "t_forwd", "cond", "xt")  if k in o}
a = run([0]); b = run(list(range(8)))
print(a.keys())
for k in range(steps):
    d = (a["x0"][k, 0].double() - b["x0"][k, 0].double()).abs()
    i = int(d.argmax())
    print(k, "x0 max %.2e at %d (val %.4f) e40031 %.2e  dR %.1e dt %.1e" % (d.max(), i, a["x0"][k, 0].flatten()[i], d.flatten()[40031],
          (a["R_forwd"][k, 0] - b["R_forwd"][k, 0]).abs().max(), (a["t_forwd"][k, 0] - b["t_forwd"][k, 0]).abs().max()))
