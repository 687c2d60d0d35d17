"""A ragged batch of real-size 3DMatch pairs (superpoint counts around 564 x 629, as `tools/bench_e2e.py` produces from 2 x 9 000 points)
through `DenoiseEngine.run_ragged`: ms per call / pairs per second for P = 1, 2, 4, 8, 16 pairs per call, with the per-family GPU time.
A secondary line (bench.py is BASELINE's N = M = 256); random-init weights, synthetic pairs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import torch
from diffreg_hip import synth, lib
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import weights, pair
DEV = "cuda:0"
variant, steps, mc = "3dmatch", 20, 200
v = synth.VARIANTS[variant]
eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                    sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV)
sizes = [(564, 629), (601, 540), (498, 655), (623, 611), (575, 590), (530, 640), (648, 602), (512, 566)]
res = {"workload": "3DMatch ragged batches, sizes cycled from %s, %d steps" % (sizes, steps), "cases": []}
for P in [int(x) for x in os.environ.get("PS", "1,2,4,8,16").split(",")]:
    prs = []
    for i in range(P):
        N, M = sizes[i % len(sizes)]
        _, p = pair(variant, N, M, 40 + i)
        prs.append({"src_feats": p["f_s"][0].to(DEV), "tgt_feats": p["f_t"][0].to(DEV), "s_pcd": p["p_s"][0].to(DEV), "t_pcd": p["p_t"][0].to(DEV),
                    "x_T": p["x_T"][0].to(DEV)})
    for _ in range(2):
        eng.run_ragged(prs, graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        eng.run_ragged(prs, graph=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    lib.prof_enable(True)
    eng.run_ragged(prs, graph=False)
    prof = lib.prof_collect()
    lib.prof_enable(False)
    fam = {k: [c, round(ms, 2)] for k, (c, ms, _) in prof.items() if c}
    res["cases"].append({"pairs_per_call": P, "ms_per_call": round(dt * 1e3, 2), "pairs_per_s": round(P / dt, 1), "family_ms": fam})
    print(res["cases"][-1], flush=True)
json.dump(res, open(os.path.join(ROOT, "profiles", "r03_ragged_real_size.json"), "w"), indent=1)
