"""Iteration sweep of the persistent Sinkhorn kernel at the roofline micro-benchmark's shape (4096 tiles of 256 x 256 float32):
time per launch against the iteration count -> the cost of one row->column dependent chain per tile, i.e. what a deeper overlap
could still hide.  Writes one JSON object (profiles/r03_sinkhorn_iteration_chain.json)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib

dev = "cuda:0"
B, N, M = 4096, 256, 256
x = torch.randn(B, N, M, device=dev) * 2
a = torch.tensor(1.0, device=dev)
out = lib.sinkhorn(x, a, 3)
byts = B * N * M * 8
rows = []
for it in (1, 2, 3, 4, 6, 8):
    for _ in range(200): lib.sinkhorn(x, a, it, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): lib.sinkhorn(x, a, it, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    rows.append(dict(iters=it, us_per_launch=us, GBps=byts / us / 1e3, frac_of_8TBps=byts / us / 1e3 / 8000))
y = torch.empty_like(x)
for _ in range(50): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): y.copy_(x)
e1.record(); torch.cuda.synchronize()
cp = e0.elapsed_time(e1) / 40 * 1e3
slope = (rows[-1]["us_per_launch"] - rows[0]["us_per_launch"]) / (rows[-1]["iters"] - rows[0]["iters"])
res = dict(shape=[B, N, M], algorithmic_bytes=byts, sweep=rows, torch_copy_same_bytes_us=cp, torch_copy_GBps=byts / cp / 1e3,
           us_per_extra_iteration=slope, us_per_iteration_per_tile_per_cu=slope / (B / 256.0),
           extrapolated_us_at_zero_iterations=rows[0]["us_per_launch"] - slope * rows[0]["iters"],
           note="one workgroup (1024 threads, the whole tile in registers) per CU: the iteration chain (row sums by DPP, column sums through "
                "LDS, two barriers) issues no memory traffic; the LDS-DMA prefetch of the next tile's first 8 rows per wave and the "
                "register loads behind the stores are what already overlaps.  The intercept is the kernel with the chain removed: compare "
                "with the tile-copy ceilings of profiles/r01_sinkhorn_copy_ceiling.txt (nt loads + nt stores: 347.6 us = 6.18 TB/s).")
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r03_sinkhorn_iteration_chain.json", "w"), indent=1)
