"""cfg3's one-sided LayerNorm launches (4 096 rows, C = 528; merge K = 576, mlp2 K = 1 024) stand-alone, k-split on / off, with the weights HOT
(the same launch repeated) and COLD (64 MB written between launches: the weights come from the memory side, as in the loop where 67 MB of layer
weights cycle through 8 x 4 MB of L2).  Per-launch HIP-event times."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
C, rows = 528, int(os.environ.get("ROWS", "4096"))
g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
lnb = lib.ln_bound(g1, b1)
ws = lib.plane_split_workspace(C, dev)
junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def make(K, resid):
    a_img, a_b = lib.planes_from_f32(torch.randn(rows, K, device=dev))
    pk = lib.pack_weight_planes(torch.randn(C, K, device=dev) / K ** 0.5, 1, C)
    o_img = torch.zeros(lib._lib.dr_plane_image_bytes(rows, C), dtype=torch.uint8, device=dev); o_b = torch.zeros(rows, device=dev)
    o32 = torch.empty(rows, C, device=dev); xr = torch.randn(rows, C, device=dev); xb = xr.abs().amax(1).contiguous()
    def f(split):
        if resid:
            lib.linear_planes(rows, C, 1, a_img, a_b, K, pk, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=xr,
                              ldr=C, bound_resid=xb, lnb=lnb, split_ws=ws if split else None)
        else:
            lib.linear_planes(rows, C, 1, a_img, a_b, K, pk, lib.PL_LN, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb,
                              split_ws=ws if split else None)
    return f
def timed(f, split, cold, n=30):
    for _ in range(5): f(split)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if cold: junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(split); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
lib.raw().dr_debug_enable_env(1)
def burst(f, split, n=40):
    """back-to-back launches between two events (no per-launch event overhead)"""
    for _ in range(5): f(split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f(split)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, K, resid in (("merge + LN (K = 576)", 576, False), ("mlp2 + LN + residual (K = 1024)", 1024, True)):
    f = make(K, resid)
    for _ in range(50): f(False)
    for split in (False, True):
        print("%-34s rows %d  k-split %-5s  hot %6.1f us   cold %6.1f us" % (name, rows, split, min(timed(f, split, False) for _ in range(3)),
                                                                            min(timed(f, split, True) for _ in range(3))))
        parts = {}
        for tag, dbg in (("whole", "0"), ("main loop", "1"), ("main loop + exchange", "64")):
            if tag.endswith("exchange") and not split:
                continue
            os.environ["DR_PG_NOEPI"] = dbg
            parts[tag] = round(min(burst(f, split) for _ in range(3)), 1)
        os.environ["DR_PG_NOEPI"] = "0"
        print("        back to back:", parts)
