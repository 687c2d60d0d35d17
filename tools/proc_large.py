"""Large-tile Procrustes timing (round 3): the top-K selection + fit of one 1024 x 2048 tile (cfg5) and of 8 tiles of 1500 x 1500
(cfg3), per value distribution.  Writes profiles/r03_procrustes_large_tiles.json."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "diff-reg_amd"))
from diffreg_hip import lib

dev = torch.device("cuda:0")
out = {"note": "dr_procrustes_f32 (stream-ordered workspace included), 50 calls after 5 warm-up, wall clock / call", "cases": []}
for P, N, M in ((1, 1024, 2048), (8, 1500, 1500), (1, 512, 512)):
    g = torch.Generator().manual_seed(1)
    base = torch.rand(P, N, M, generator=g)
    ps, pt = torch.rand(P, N, 3, generator=g).to(dev), torch.rand(P, M, 3, generator=g).to(dev)
    sm, tm = torch.ones(P, N, dtype=torch.bool, device=dev), torch.ones(P, M, dtype=torch.bool, device=dev)
    for kind in ("distinct", "quantised16", "flat", "peaked"):
        c = base
        if kind == "quantised16":
            c = (base * 16).floor() / 16
        elif kind == "flat":
            c = torch.full_like(base, 1.0 / M)
        elif kind == "peaked":
            c = base ** 64
        c = c.to(dev)
        for _ in range(5):
            lib.procrustes(c, ps, pt, sm, tm, 1.0, 1e9)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            lib.procrustes(c, ps, pt, sm, tm, 1.0, 1e9)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        out["cases"].append({"P": P, "N": N, "M": M, "values": kind, "us_per_call": round(us, 1)})
        print(out["cases"][-1], flush=True)
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r03_procrustes_large_tiles.json"), "w"), indent=1)
