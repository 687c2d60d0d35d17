"""Collate-time ops on a 3DMatch-sized pair of clouds: the device level loop (grid subsampling + radius
neighbours for the 4 KPFCN levels) against the reference's own C++ (oracle/_ref, kind = "reference") on one host core, which is
how a data-loader worker runs it (a secondary line; bench.py is the denoising loop)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth, lib
from diffreg_hip.collate import build_kpfcn_inputs
from oracle import collate_oracle as co

n = int(os.environ.get("NPTS", "30000"))
rng = np.random.default_rng(0)
def cloud(n, seed):
    # a folded sheet in a 4 x 3 x 2.5 m room (surface-like, as a depth scan), 2.5 cm pre-voxelised like 3DMatch
    u = synth.hash_uniform(seed, 1, (n * 3, 3), 0.0, 1.0)
    P = np.stack([4 * u[:, 0], 3 * u[:, 1], 1.2 + 0.6 * np.sin(3 * u[:, 0]) * np.cos(2 * u[:, 1]) + 0.02 * u[:, 2]], 1).astype(np.float32)
    # (3DMatch clouds come pre-voxelised at 2.5 cm: done with the device op; the reference C++ only serves the cpu_baseline leg)
    from diffreg_hip.collate import batch_grid_subsampling_kpconv
    sp, _ = batch_grid_subsampling_kpconv(torch.from_numpy(P).cuda(), torch.tensor([len(P)], dtype=torch.int32).cuda(), sampleDl=0.025)
    return sp[:n].cpu().numpy()
A, B = cloud(n, 1), cloud(n, 2)
P = np.concatenate([A, B]); L = np.array([len(A), len(B)], np.int32)
cfg = dict(architecture=synth.KPFCN_ARCH, first_subsampling_dl=0.025, conv_radius=2.5, deform_radius=5.0)
limits = [38, 36, 36, 38]
Pd, Ld = torch.from_numpy(P).cuda(), torch.from_numpy(L).cuda()
for _ in range(3):
    out = build_kpfcn_inputs(Pd, Ld, cfg, limits)
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 10
for _ in range(reps):
    out = build_kpfcn_inputs(Pd, Ld, cfg, limits)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / reps * 1e3
res = {"points_per_level": [int(p.shape[0]) for p in out["points"]], "neighbor_widths": [int(x.shape[1]) for x in out["neighbors"]],
       "gpu_ms_per_pair_level_loop": gpu_ms}
# kernel-only time of the two largest calls (HIP events)
def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
res["level0_conv_neighbors_ms"] = timed(lambda: lib.radius_neighbors(Pd, Pd, Ld, Ld, 0.0625, limits[0]))
res["level0_grid_subsample_ms"] = timed(lambda: lib.grid_subsample(Pd, Ld, 0.05))
if co.ref_lib() is not None and os.environ.get("CPU", "1") == "1":
    t0 = time.perf_counter()
    pts, lens, r = P, L, 0.025 * 2.5
    for lvl in range(4):
        conv = co.ref_batch_query(pts, pts, lens, lens, r)[:, :limits[lvl]]
        if lvl == 3:
            break
        pp, pl = co.ref_subsample_batch(pts, lens, 2 * r / 2.5)
        pool = co.ref_batch_query(pp, pts, pl, lens, r)[:, :limits[lvl]]
        up = co.ref_batch_query(pts, pp, lens, pl, 2 * r)[:, :limits[lvl]]
        pts, lens, r = pp, pl, 2 * r
    res["cpu_baseline"] = {"value": (time.perf_counter() - t0) * 1e3, "unit": "ms per pair (level loop)", "cores": 1, "kind": "reference",
                           "sample": "one pair of %d + %d points, oracle/_ref (the reference's C++ compiled with g++ -O2)" % (len(A), len(B))}
print(json.dumps(res))
