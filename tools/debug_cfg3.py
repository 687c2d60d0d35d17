"""per-step deviation of the 4D 512^2 20-step loop (pair of the reference golden) from the reference vectors"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import T, weights, pair
DEV = "cuda:0"
variant, N, M, steps, mc = "4dmatch", 512, 512, 20, 40
v = synth.VARIANTS[variant]
g = np.load(os.path.join(ROOT, "tests/golden/4dmatch_loop_n512_s20_mc40_masked.npz"))
for strict in (False, True):
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                        sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, strict_f64=strict, planes=True)
    _, p = pair(variant, N, M, 62)
    ms = (torch.arange(N)[None] < 470).to(DEV); mt = (torch.arange(M)[None] < 391).to(DEV)
    noise = T(synth.step_noise(N, M, 62, steps))[:, None].to(DEV)
    out = eng.run(p["f_s"].to(DEV), p["f_t"].to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), p["x_T"].to(DEV), ms, mt, noise=noise, trace=True)
    Rf = out["R_forwd"][:, 0].cpu().numpy(); tf = out["t_forwd"][:, 0].cpu().numpy()
    x0 = out["x0"][:, 0].cpu().numpy()
    print("strict", strict)
    for k in range(steps):
        print(k, "dR %.2e dt %.2e cond %.3f/%.3f x0corner %.2e x0sum %.6f/%.6f" % (np.abs(Rf[k] - g["R_forwd"][k]).max(), np.abs(tf[k] - g["t_forwd"][k]).max(),
              float(out["cond"][k, 0]), float(g["cond"][k]), np.abs(x0[k][:16, :16] - g["x0_corner"][k]).max(), x0[k].astype(np.float64).sum(), g["x0_sum"][k]))
