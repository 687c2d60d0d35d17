"""Large-tile Sinkhorn (the shapes beyond the register-resident kernel: co-resident form where it fits, grid form beyond): graph-replayed time per call.
cfg5: 1 x 1024 x 2048 (2D-3D), cfg3: 8 x 512 x 512 (4DMatch), a real 3DMatch pair: 1 x 564 x 629.  Writes gpurun_out/r03_sinkhorn_large_tiles.json."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib

dev = "cuda:0"
res = []
for (B, N, M) in ((1, 1024, 2048), (8, 512, 512), (1, 564, 629), (4, 1024, 2048)):
    x = torch.randn(B, N, M, device=dev) * 2
    a = torch.tensor(1.0, device=dev)
    sm = torch.ones(B, N, dtype=torch.bool, device=dev); tm = torch.ones(B, M, dtype=torch.bool, device=dev)
    tm[:, M - 7:] = False
    out = lib.sinkhorn(x, a, 3, sm, tm, apply_mask=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): lib.sinkhorn(x, a, 3, sm, tm, apply_mask=True, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): lib.sinkhorn(x, a, 3, sm, tm, apply_mask=True, out=out)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    byts = B * N * M * 8
    res.append(dict(shape=[B, N, M], us_per_call=us, algorithmic_bytes=byts, GBps=byts / us / 1e3, frac_of_8TBps=byts / us / 1e3 / 8000))
    print(res[-1])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/r03_sinkhorn_large_tiles.json", "w"), indent=1)
