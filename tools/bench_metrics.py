"""Evaluation harness on P synthetic pairs: time per call of each kernel family on the GPU (HIP events on the
launch stream) and of the oracle on the host (a secondary line; bench.py is the denoising loop).
RANSAC work unit: one hypothesis x correspondence evaluation = 27 fp64 flop (9 FMA + 3 sub for R s + t - y, 3 for |.|^2,
compare + accumulate)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import lib, synth
from tests.helpers import metrics_scene

P = int(os.environ.get("PAIRS", "64"))
N = M = int(os.environ.get("NM", "256"))
ITERS = int(os.environ.get("ITERS", "50000"))
scs = [metrics_scene(N, M, 100 + p) for p in range(P)]
cap = N + M
matches = torch.zeros(P, cap, 3, dtype=torch.int64)
count = torch.zeros(P, dtype=torch.int32)
for p, sc in enumerate(scs):
    k = min(len(sc["matches"]), cap)
    matches[p, :k] = sc["matches"][:k]; count[p] = k
cat = lambda key: torch.cat([sc[key] for sc in scs]).cuda()
matches, count = matches.cuda(), count.cuda()
s_pcd, t_pcd, t_pcd4, rot, trn = cat("s_pcd"), cat("t_pcd"), cat("t_pcd4"), cat("rot"), cat("trn")
info = torch.stack([torch.from_numpy(sc["info"]) for sc in scs]).cuda()
raw, flow = torch.cat([sc["raw_pcd"] for sc in scs]).cuda(), torch.cat([sc["raw_flow"] for sc in scs]).cuda()
midx = torch.cat([sc["metric_index"] for sc in scs]).cuda()
roff = torch.tensor(np.cumsum([0] + [len(sc["raw_pcd"]) for sc in scs]), dtype=torch.int32).cuda()
qoff = torch.tensor(np.cumsum([0] + [len(sc["metric_index"]) for sc in scs]), dtype=torch.int32).cuda()
maxq = max(len(sc["metric_index"]) for sc in scs)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {"pairs": P, "N": N, "M": M, "mean_matches": float(count.float().mean()), "ransac_iters": ITERS}
res["inlier_ratio_ms"] = timed(lambda: lib.inlier_ratio(matches, count, s_pcd, t_pcd, rot, trn, 0.1))
res["nrfmr_ms"] = timed(lambda: lib.nrfmr(matches, count, s_pcd, t_pcd4, raw, flow, roff, midx, qoff, maxq, rot, trn))
rs = {}
def run_ransac():
    rs.update(lib.ransac_corr(matches, count, s_pcd, t_pcd, 0.05, ITERS, seed=0))
res["ransac_ms"] = timed(run_ransac, reps=10, warm=2)
res["recall_ms"] = timed(lambda: lib.registration_recall(rs["rot"], rs["trn"], rot, trn, info, 0.2))
evals = float(count.double().sum()) * ITERS
res["ransac_gevals_per_s"] = evals / res["ransac_ms"] / 1e6
res["ransac_fp64_tflops"] = evals * 27 / res["ransac_ms"] / 1e9
res["ransac_pairs_per_s"] = P / res["ransac_ms"] * 1e3
err, ok = lib.registration_recall(rs["rot"], rs["trn"], rot, trn, info, 0.2)
res["registration_recall"] = float(ok.float().mean())
res["mean_fitness"] = float(rs["fitness"].mean())
if os.environ.get("CPU", "1") == "1":
    from oracle import metrics_oracle as mo
    sc = scs[0]
    t0 = time.perf_counter()
    o = mo.ransac_corr(sc["s_pcd"][0].numpy(), sc["t_pcd"][0].numpy(), sc["matches"][:, 1:].numpy(), 0.05, ITERS, seed=0, pair_id=0)
    res["cpu_oracle_ransac_s_per_pair"] = time.perf_counter() - t0
    res["ransac_R_max_abs_diff_vs_oracle"] = float(np.abs(rs["rot"][0].cpu().numpy() - o["R"]).max())
    t0 = time.perf_counter()
    mo.nrfmr(sc["matches"], sc["s_pcd"], sc["t_pcd4"], [sc["raw_pcd"]], [sc["raw_flow"]], [sc["metric_index"]], sc["rot"], sc["trn"])
    mo.inlier_ratio(sc["matches"], sc["s_pcd"], sc["t_pcd"], sc["rot"], sc["trn"], 0.1)
    res["cpu_oracle_ir_nrfmr_ms_per_pair"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(res))
