"""2D-3D variant (BASELINE configs[4]: N = 1024 point nodes x M = 2048 image patches, 10 denoise steps): seconds per
pair of dr_denoise_loop_2d3d on the GPU and of the oracle on the host (a secondary line; bench.py is configs[1])."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine2D3D
N, M, steps, mc = 1024, 2048, 10, 200
P = int(os.environ.get("P", "1"))
Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
W = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in Wn.items()}
eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device="cuda:0")
prs = [synth.make_pair_2d3d(N, M, 60 + i, weights=Wn) for i in range(P)]
d = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).cuda()
args = [d(k) for k in ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")]
for _ in range(2):
    out = eng.run(*args)
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    out = eng.run(*args)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / reps
res = {"workload": "2D-3D N=%d x M=%d, %d steps, %d pair(s) per call" % (N, M, steps, P), "gpu_s_per_call": gpu, "gpu_pairs_per_s": P / gpu}
if os.environ.get("CPU", "1") == "1":
    from oracle import diffreg_oracle as orc
    torch.set_num_threads(16)
    p = prs[0]; q = lambda k: torch.from_numpy(p[k])[None]
    ms = torch.ones(1, N, dtype=torch.bool); mt = torch.ones(1, M, dtype=torch.bool)
    t0 = time.perf_counter()
    orc.denoise_loop_2d3d(W, synth.VARIANTS["2d3d"], q("img_feats"), q("img_dino"), q("img_pixels"), q("pcd_feats"), q("s_pcd"), q("t_pcd_da"),
                          ms, mt, mt, q("x_T"), steps, mc)
    res["cpu_oracle_s_per_pair_16_threads"] = time.perf_counter() - t0
from diffreg_hip import lib
lib.prof_enable(True)
eng.run(*args)
prof = lib.prof_collect()
lib.prof_enable(False)
tot = sum(v[1] for v in prof.values())
res["kernel_families"] = {k: {"launches": v[0], "ms": v[1], "share": v[1] / tot} for k, v in prof.items() if v[0]}
dom = max(prof, key=lambda k: prof[k][1])
c, ms_, work = prof[dom]
res["dominant_family"] = {"family": dom, "avg_us_per_launch": ms_ / c * 1e3, "achieved": work / (ms_ * 1e-3) / (1e9 if dom in ("sinkhorn", "state", "position_code", "layernorm") else 1e12),
                          "unit": "GB/s" if dom in ("sinkhorn", "state", "position_code", "layernorm") else "TFLOP/s"}
print(json.dumps(res))
