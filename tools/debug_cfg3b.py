"""batch of 8 pairs at 512^2 (4D) vs the pairs' own B = 1 runs, step by step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import T, weights, pair
DEV = "cuda:0"
variant, N, M, mc = "4dmatch", 512, 512, 40
steps = int(os.environ.get("STEPS", "20"))
v = synth.VARIANTS[variant]
eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                    sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, planes=True)
cases = [(470, 391, 62), (512, 512, 61), (500, 480, 63), (512, 300, 64), (333, 512, 65), (450, 450, 66), (512, 511, 67), (400, 390, 68)][:int(os.environ.get("P", "8"))]
prs = [pair(variant, N, M, c[2])[1] for c in cases]
cat = lambda k: torch.cat([q[k] for q in prs]).to(DEV)
ms = torch.stack([torch.arange(N) < c[0] for c in cases]); mt = torch.stack([torch.arange(M) < c[1] for c in cases])
noise = torch.stack([T(synth.step_noise(N, M, c[2], steps)) for c in cases], 1)
out = eng.run(cat("f_s"), cat("f_t"), cat("p_s"), cat("p_t"), cat("x_T"), ms.to(DEV), mt.to(DEV), noise=noise.to(DEV), trace=True)
B = {k: out[k].cpu().clone() for k in ("R_forwd", "x0", "cond", "conf_matrix_pred")}
for i, q in enumerate(prs):
    one = eng.run(q["f_s"].to(DEV), q["f_t"].to(DEV), q["p_s"].to(DEV), q["p_t"].to(DEV), q["x_T"].to(DEV), ms[i:i + 1].to(DEV), mt[i:i + 1].to(DEV),
                  noise=noise[:, i:i + 1].to(DEV), trace=True)
    dR = (one["R_forwd"][:, 0].cpu() - B["R_forwd"][:, i]).abs().amax((1, 2)).numpy()
    dx = (one["x0"][:, 0].cpu() - B["x0"][:, i]).abs().amax((1, 2)).numpy()
    print("pair", i, "dR", " ".join("%.0e" % a for a in dR))
    print("       dx0", " ".join("%.0e" % a for a in dx), "cond", " ".join("%.1f" % a for a in one["cond"][:, 0].cpu().numpy()))
