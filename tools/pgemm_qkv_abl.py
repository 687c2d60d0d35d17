"""timing of the 3-block F32 launch under the DR_PG_ABL ablations (DR_PG_STAMPS=1 builds)"""
import os, sys
os.environ["DR_PG_STAMPS"] = "1"
os.environ["DR_DIAGNOSTICS"] = "1"     # the library reads DR_* variables only under this switch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
import torch
from diffreg_hip import lib
dev = torch.device("cuda:0")
rows, C = 32768, 432
x = torch.randn(rows, C, device=dev)
img, bnd = lib.planes_from_f32(x)
for nblk in (1, 2, 3):
    pk = lib.pack_weight_planes(torch.randn(nblk * C, C, device=dev) / C ** 0.5, nblk, C)
    out = torch.empty(nblk, rows, C, device=dev)
    f = lambda: lib.linear_planes(rows, C, nblk, img, bnd, C, pk, lib.PL_F32, out=out, ldo=C, blk_stride=rows * C)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("ABL", os.environ.get("DR_PG_ABL", "0"), "nblk", nblk, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
