"""BASELINE configs[2] shape (4DMatch: N = M = 512, C = 528, 20 steps, 8 pairs per call): seconds per call on the GPU
(a secondary line; bench.py is configs[1])."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
variant, N, M, steps, mc = "4dmatch", 512, 512, 20, 40.0
P = int(os.environ.get("P", "8"))
v = synth.VARIANTS[variant]
W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=24.0).items()}
eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                    sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device="cuda:0")
prs = [synth.make_pair(N, M, v["C"], seed=300 + i) for i in range(P)]
st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).cuda()
noise = torch.from_numpy(np.stack([synth.step_noise(N, M, 300 + i, steps) for i in range(P)], 1)).cuda()
ms = torch.ones(P, N, dtype=torch.bool, device="cuda"); mt = torch.ones(P, M, dtype=torch.bool, device="cuda")
run = lambda: eng.run(st_fs, st_ft, st_ps, st_pt, st_x, ms, mt, noise=noise, graph=os.environ.get("GRAPH", "1") == "1")
st_fs, st_ft, st_ps, st_pt, st_x = st("src_feats"), st("tgt_feats"), st("s_pcd"), st("t_pcd"), st("x_T")
for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": "4DMatch N=M=512, C=528, %d steps, %d pairs per call" % (steps, P), "gpu_s_per_call": dt, "gpu_pairs_per_s": P / dt}))
