"""Batched Sinkhorn micro-benchmark: GB/s of dr_sinkhorn_f32 vs the 8 TB/s HBM peak.
algorithmic bytes per tile = N*M*(sizeof in + sizeof out)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib  # noqa: E402


def run(B, N, M, dtype=torch.float32, reps=20, **kw):
    dev = "cuda:0"
    x = torch.randn(B, N, M, device=dev, dtype=dtype) * 2
    a = torch.tensor(1.0, device=dev)
    out = lib.sinkhorn(x, a, 3, **kw)
    for _ in range(3):
        lib.sinkhorn(x, a, 3, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.sinkhorn(x, a, 3, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = B * N * M * (x.element_size() + out.element_size())
    return dict(B=B, N=N, M=M, dtype=str(dtype).split(".")[-1], us=ms * 1e3, GBps=byts / ms / 1e6,
                frac_of_8TBps=byts / ms / 1e6 / 8000.0, **{k: v for k, v in kw.items() if k != "out"})


if __name__ == "__main__":
    rows = []
    for B in (1, 8, 64, 256, 1024, 4096):
        rows.append(run(B, 256, 256))
    rows.append(run(4096, 128, 128))
    rows.append(run(2048, 256, 256, torch.float64))
    rows.append(run(2048, 256, 256, torch.float64, out_f32=True))
    rows.append(run(64, 512, 512))
    rows.append(run(256, 256, 256, torch.float64, strict=True))
    # device copy for context (same bytes as 4096 tiles)
    x = torch.empty(4096 * 256 * 256, device="cuda:0"); y = torch.empty_like(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): y.copy_(x)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rows.append(dict(copy_GBps=2 * x.numel() * 4 / ms / 1e6))
    for r in rows:
        print(json.dumps(r))
