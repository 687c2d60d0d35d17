"""GPU check + timing of the plane-image GEMM (dr_linear_planes_f32) against an fp64 product.
    python tools/pgemm_check.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
import torch
from diffreg_hip import lib

dev = torch.device("cuda:0")
torch.manual_seed(1)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
C = 432


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


# ---- image round trip
x = torch.randn(rows, C, device=dev) * torch.rand(rows, 1, device=dev) * 5
img, bnd = lib.planes_from_f32(x)
back = lib.planes_to_f32(img, bnd, rows, C)
print("image round trip rel err", rel(back, x.double()), "bound ok", bool((bnd >= x.abs().amax(1)).all()))

# ---- F32 mode, 3 blocks with rotary on blocks 0, 1
W = torch.randn(3 * C, C, device=dev) / C ** 0.5
pk = lib.pack_weight_planes(W, 3, C)
ang = torch.rand(rows, C // 2, device=dev) * 6.28
cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
out = torch.full((rows, 3 * C), float("nan"), device=dev)
lib.linear_planes(rows, C, 3, img, bnd, C, pk, lib.PL_F32, out=out, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C, scale=0.5)
ref = x.double() @ W.double().t()
def rot(z, c, s):
    z = z.clone()
    e, o = z[:, 0::2], z[:, 1::2]
    return torch.stack([e * c - o * s, o * c + e * s], -1).reshape(z.shape)
refr = torch.cat([rot(ref[:, :C], cosT.double(), sinT.double()), rot(ref[:, C:2 * C], cosT.double(), sinT.double()), ref[:, 2 * C:]], 1) * 0.5
print("F32 mode (q|k|v + rotary) rel err", rel(out, refr), "nan", int(torch.isnan(out).sum()))
e32 = rel((x @ W.t()), ref)
print("   torch fp32 matmul rel err", e32)

# ---- LN mode with residual, then PLANES mode on [x | msg], then LN on hid
g1, b1 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
Wm = torch.randn(C, C, device=dev) / C ** 0.5
pkm = lib.pack_weight_planes(Wm, 1, C)
lnb = lib.ln_bound(g1, b1)
msg_img = torch.zeros_like(img); msg_b = torch.zeros(rows, device=dev)
msg32 = torch.full((rows, C), float("nan"), device=dev)
lib.linear_planes(rows, C, 1, img, bnd, C, pkm, lib.PL_LN, out=msg32, ldo=C, out_image=msg_img, out_image_k=C, out_k0=0, out_bound=msg_b,
                  gamma=g1, beta=b1, lnb=lnb)
refm = torch.nn.functional.layer_norm(x.double() @ Wm.double().t(), (C,), g1.double(), b1.double())
print("LN mode rel err fp32 out", rel(msg32, refm), " plane out", rel(lib.planes_to_f32(msg_img, msg_b, rows, C), refm),
      "bound ok", bool((msg_b >= refm.abs().amax(1).float()).all()))
W1 = torch.randn(2 * C, 2 * C, device=dev) / (2 * C) ** 0.5
pk1 = lib.pack_weight_planes(W1, 2, C)
hid_img = torch.zeros(lib.raw().dr_plane_image_bytes(rows, 2 * C), dtype=torch.uint8, device=dev); hid_b = torch.zeros(rows, device=dev)
lib.linear_planes(rows, C, 2, img, bnd, C, pk1, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=hid_img, out_image_k=2 * C, out_k0=0,
                  out_bound=hid_b, relu=True)
refh = torch.relu(torch.cat([x.double(), refm], 1) @ W1.double().t())
hid = lib.planes_to_f32(hid_img, hid_b, rows, 2 * C)
print("PLANES mode ([x|msg] -> relu hid) rel err", rel(hid, refh), "bound ok", bool((hid_b >= refh.abs().amax(1).float()).all()))
W2 = torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5
pk2 = lib.pack_weight_planes(W2, 1, C)
o32 = torch.full((rows, C), float("nan"), device=dev); o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev)
lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_k0=0, out_bound=o_b,
                  gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb)
refo = x.double() + torch.nn.functional.layer_norm(refh @ W2.double().t(), (C,), g1.double(), b1.double())
print("LN+residual mode rel err fp32", rel(o32, refo), "plane", rel(lib.planes_to_f32(o_img, o_b, rows, C), refo),
      "bound ok", bool((o_b >= refo.abs().amax(1).float()).all()))

# ---- merge with head-padded k order (4 heads of 108 -> 112)
att = torch.randn(rows, C, device=dev)
attp = torch.zeros(rows, 448, device=dev)
for hh in range(4):
    attp[:, 112 * hh:112 * hh + 108] = att[:, 108 * hh:108 * (hh + 1)]
aimg, ab = lib.planes_from_f32(attp)
pkp = lib.pack_weight_planes(Wm, 1, C, piece_len=108, piece_pad=112)
o2 = torch.full((rows, C), float("nan"), device=dev)
lib.linear_planes(rows, C, 1, aimg, ab, 448, pkp, lib.PL_F32, out=o2, ldo=C, blk_stride=0)
print("head-padded k order rel err", rel(o2, att.double() @ Wm.double().t()))

# ---- timing
if rows >= 4096:
    def timeit(f, n=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = timeit(lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pkm, lib.PL_F32, out=o2, ldo=C))
    print("F32 432x432: %.1f us, %.1f TFLOP/s" % (t, 2.0 * rows * C * C / t / 1e6))
    t = timeit(lambda: lib.linear_planes(rows, C, 3, img, bnd, C, pk, lib.PL_F32, out=out, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C))
    out3 = torch.empty(3, rows, C, device=dev)
    t = timeit(lambda: lib.linear_planes(rows, C, 3, img, bnd, C, pk, lib.PL_F32, out=out3, ldo=C, blk_stride=rows * C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C))
    print("F32 q|k|v as three [T,C] matrices: %.1f us" % t)
    print("F32 q|k|v 432x1296: %.1f us, %.1f TFLOP/s" % (t, 2.0 * rows * C * 3 * C / t / 1e6))
    t = timeit(lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pkm, lib.PL_LN, out_image=msg_img, out_image_k=C, out_bound=msg_b, gamma=g1, beta=b1, lnb=lnb))
    print("merge+LN -> planes: %.1f us, %.1f TFLOP/s" % (t, 2.0 * rows * C * C / t / 1e6))
    t = timeit(lambda: lib.linear_planes(rows, C, 2, img, bnd, C, pk1, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=hid_img, out_image_k=2 * C, out_bound=hid_b, relu=True))
    print("mlp0 864x864 -> planes: %.1f us, %.1f TFLOP/s" % (t, 2.0 * rows * 4 * C * C / t / 1e6))
    t = timeit(lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb))
    print("mlp2+LN+res 864x432: %.1f us, %.1f TFLOP/s" % (t, 2.0 * rows * 2 * C * C / t / 1e6))
