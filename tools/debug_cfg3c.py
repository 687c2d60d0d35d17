"""largest non-exempt deviations of the 4D 512^2 loop's x0_last / conf from the reference vectors (B = 1, plane path)"""
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
fx = "4dmatch_loop_n512_s20_mc40_masked"
g = np.load(os.path.join(ROOT, "tests/golden/%s.npz" % fx))
ex = json.load(open(os.path.join(ROOT, "tests/golden/loop_exemptions.json")))["fixtures"][fx]
for planes in (True, False):
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                        sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, planes=planes)
    _, p = pair(variant, N, M, 62)
    ms = (torch.arange(N)[None] < 470).to(DEV); mt = (torch.arange(M)[None] < 391).to(DEV)
    noise = T(synth.step_noise(N, M, 62, steps))[:, None].to(DEV)
    out = eng.run(p["f_s"].to(DEV), p["f_t"].to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), p["x_T"].to(DEV), ms, mt, noise=noise, trace=True)
    for key, got in (("x0_last", out["x0"][-1, 0].cpu().numpy()), ("conf", out["conf_matrix_pred"][0].cpu().numpy())):
        idx = np.asarray(ex[key]["index"], dtype=np.int64)
        ref = g[key].astype(np.float64).ravel(); gotf = got.astype(np.float64).ravel()
        dev = np.abs(gotf - ref); dev[idx] = 0
        top = np.argsort(-dev)[:6]
        print("planes", planes, key, [(int(i), "%.2e" % dev[i], "%.5f" % ref[i]) for i in top])
# the batch of 8 of tests/test_loop_gpu.py::test_cfg3_4dmatch_512_batch8_20_steps, pair 0
cases = [(470, 391, 62), (512, 512, 61), (500, 480, 63), (512, 300, 64), (333, 512, 65), (450, 450, 66), (512, 511, 69), (400, 390, 68)]
prs = [pair(variant, N, M, c[2])[1] for c in cases]
cat = lambda k: torch.cat([q[k] for q in prs]).to(DEV)
ms = torch.stack([torch.arange(N) < c[0] for c in cases]); mt = torch.stack([torch.arange(M) < c[1] for c in cases])
noise = torch.stack([T(synth.step_noise(N, M, c[2], steps)) for c in cases], 1)
eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                    sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, planes=True)
out = eng.run(cat("f_s"), cat("f_t"), cat("p_s"), cat("p_t"), cat("x_T"), ms.to(DEV), mt.to(DEV), noise=noise.to(DEV), trace=True)
for key, got in (("x0_last", out["x0"][-1, 0].cpu().numpy()), ("conf", out["conf_matrix_pred"][0].cpu().numpy())):
    idx = np.asarray(ex[key]["index"], dtype=np.int64)
    ref = g[key].astype(np.float64).ravel(); gotf = got.astype(np.float64).ravel()
    dev = np.abs(gotf - ref); dev[idx] = 0
    top = np.argsort(-dev)[:6]
    print("batch8 pair0", key, [(int(i), "%.2e" % dev[i], "%.5f" % ref[i]) for i in top])
Rf = out["R_forwd"][:, 0].cpu().numpy()
print("dR per step", ["%.1e" % np.abs(Rf[k] - g["R_forwd"][k]).max() for k in range(steps)])
