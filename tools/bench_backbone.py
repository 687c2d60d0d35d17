"""KPFCN coarse phase on a larger synthetic stacked cloud: seconds per forward on the GPU and of the
oracle on the host (a secondary line; bench.py is the denoising loop)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.backbone import KPFCNEngine
n = int(os.environ.get("NPTS", "12000"))
g = np.load(os.path.join(ROOT, "tests", "golden", "kpfcn_coarse.npz"))
kp = {k[3:]: g[k] for k in g.files if k.startswith("kp:")}
sd = {k: torch.from_numpy(v) for k, v in synth.make_kpfcn_weights(kp).items()}
t0 = time.perf_counter()
b = synth.make_kpfcn_batch(n_src=n, n_tgt=n, seed=1, limit=(35, 35, 35, 35), extent=(n / 1400.0) ** 0.5)
tb = dict(points=[torch.from_numpy(p) for p in b["points"]], neighbors=[torch.from_numpy(p) for p in b["neighbors"]],
          pools=[torch.from_numpy(p) for p in b["pools"]], upsamples=[torch.from_numpy(p) for p in b["upsamples"]],
          features=torch.from_numpy(b["features"]))
t_collate = time.perf_counter() - t0
eng = KPFCNEngine(sd, device="cuda:0")
db = {k: [t.cuda() for t in v] if isinstance(v, list) else v.cuda() for k, v in tb.items()}
for _ in range(2):
    out = eng.forward(db)
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 10
for _ in range(reps):
    out = eng.forward(db)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / reps
res = {"points_per_layer": [len(p) for p in b["points"]], "gpu_ms_per_forward": gpu * 1e3, "numpy_brute_force_collate_s": t_collate}
if os.environ.get("CPU", "1") == "1":
    from oracle import kpfcn_oracle as ko
    torch.set_num_threads(16)
    t0 = time.perf_counter()
    ref = ko.kpfcn_coarse(sd, tb)
    res["cpu_oracle_ms_16_threads"] = (time.perf_counter() - t0) * 1e3
    res["max_abs_diff_vs_oracle"] = float((out.cpu() - ref).abs().max())
print(json.dumps(res))
