"""Raw clouds -> matches -> pose, every stage on the device (rows f4, f1, a1-a8, f2 of SURVEY section 8), one pair at a time as
the reference's tester runs it (B = 1): stage latencies on MI355X and the reference / oracle stage on the host beside them.
A secondary line (bench.py is the denoising loop at BASELINE's N = M = 256).  Random-init weights, synthetic clouds."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth, metrics
from diffreg_hip.backbone import KPFCNEngine
from diffreg_hip.collate import collate_fn_device
from diffreg_hip.engine import DenoiseEngine
from oracle import collate_oracle as co

n = int(os.environ.get("NPTS", "9000"))
def cloud(n, seed, R=None, t=None):
    u = synth.hash_uniform(seed, 1, (n * 3, 3), 0.0, 1.0)
    P = np.stack([2.2 * u[:, 0], 1.7 * u[:, 1], 1.2 + 0.5 * np.sin(3 * u[:, 0]) * np.cos(2 * u[:, 1]) + 0.02 * u[:, 2]], 1)
    if R is not None:
        P = P @ R.T + t
    # (3DMatch clouds come pre-voxelised at 2.5 cm: done with the device op; the oracle only serves the cpu_baseline leg)
    from diffreg_hip.collate import batch_grid_subsampling_kpconv
    sp, _ = batch_grid_subsampling_kpconv(torch.from_numpy(P.astype(np.float32)).cuda(), torch.tensor([len(P)], dtype=torch.int32).cuda(), sampleDl=0.025)
    return sp[:n].cpu().numpy()
R = synth._rodrigues(np.array([0.2, 0.1, 1.0]), 0.3); t = np.array([0.1, -0.05, 0.02])
A, B = cloud(n, 1), cloud(n, 1, R, t)                      # the same surface seen from a second pose
g = np.load(os.path.join(ROOT, "tests", "golden", "kpfcn_coarse.npz"))
kp = {k[3:]: g[k] for k in g.files if k.startswith("kp:")}
bsd = {k: torch.from_numpy(v) for k, v in synth.make_kpfcn_weights(kp).items()}
v = synth.VARIANTS["3dmatch"]
W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=24.0).items()}
kc = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), deform_radius=5.0)
limits = [38, 36, 36, 38]
dev = "cuda:0"
bb = KPFCNEngine(bsd, device=dev)
loop = DenoiseEngine(W, variant="3dmatch", C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=20, sk_iters=3,
                     sample_rate=1.0, max_condition_num=200.0, device=dev)
Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
rot = torch.tensor(R, dtype=torch.float32).cuda(); trn = torch.tensor(t, dtype=torch.float32).cuda()

def stage_times(reps):
    ts = dict(collate=0.0, backbone=0.0, loop=0.0, harness=0.0)
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d = collate_fn_device([(Ad, Bd, rot, trn)], kc, limits)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        feats = bb.forward(d)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ns, nt = [int(x) for x in d["stack_lengths"][-2]]
        pts = d["points"][-2]
        x_T = torch.randn(1, ns, nt, device=dev)
        out = loop.run(feats[None, :ns].contiguous(), feats[None, ns:].contiguous(), pts[None, :ns].contiguous(), pts[None, ns:].contiguous(),
                       x_T, graph=True)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        ev = metrics.evaluate_pairs(out["matches_padded"], out["match_count"], pts[None, :ns].contiguous(), pts[None, ns:].contiguous(),
                                    rot[None], trn[None])
        torch.cuda.synchronize(); t4 = time.perf_counter()
        for k, dt in zip(ts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            ts[k] += dt * 1e3 / reps
    return ts, (ns, nt), d

stage_times(3)
ts, (ns, nt), d = stage_times(10)
res = {"raw_points": [len(A), len(B)], "points_per_level": [int(p.shape[0]) for p in d["points"]], "coarse_N_M": [ns, nt],
       "gpu_ms_per_pair": ts, "gpu_ms_per_pair_total": sum(ts.values()), "pairs_per_s_latency_mode": 1e3 / sum(ts.values())}
if co.ref_lib() is not None and os.environ.get("CPU", "1") == "1":
    P = np.concatenate([A, B]); L = np.array([len(A), len(B)], np.int32)
    t0 = time.perf_counter()
    pts, lens, r = P, L, 0.025 * 2.5
    for lvl in range(4):
        co.ref_batch_query(pts, pts, lens, lens, r)
        if lvl == 3:
            break
        pp, pl = co.ref_subsample_batch(pts, lens, 2 * r / 2.5)
        co.ref_batch_query(pp, pts, pl, lens, r); co.ref_batch_query(pts, pp, lens, pl, 2 * r)
        pts, lens, r = pp, pl, 2 * r
    res["cpu_collate_ms_reference_cpp_1_core"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(res))
