"""ABLATION (debug knob DR_PG_NOEPI, bits: 1 no epilogue, 4 no fp32 row stores, 8 no image stores, 16 no residual loads, 32 no rotary table loads): what the
epilogue of each layer GEMM shape costs per launch at 65 536 rows, and which of its memory streams it is.  Results are wrong by construction; only times matter."""
import os, sys, torch
os.environ["DR_DIAGNOSTICS"] = "1"; os.environ["DR_PG_HALF"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
C, KP, rows = 432, 432, int(os.environ.get("ROWS", "65536"))
x = torch.randn(rows, KP, device=dev)
img, bnd = lib.planes_from_f32(x)
g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
lnb = lib.ln_bound(g1, b1)
msg_img, msg_b = lib.planes_from_f32(torch.randn(rows, KP, device=dev))
hid_img, hid_b = lib.planes_from_f32(torch.randn(rows, 2 * C, device=dev))
o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, 3 * C, device=dev)
h_img = torch.zeros_like(hid_img); h_b = torch.zeros(rows, device=dev)
q_img = torch.zeros(3 * (img.numel() // KP * 448), dtype=img.dtype, device=dev); q_b = torch.zeros(3 * rows, device=dev)
xr = torch.randn(rows, C, device=dev)
ang = torch.rand(rows, C // 2, device=dev); cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
pk3 = lib.pack_weight_planes(torch.randn(3 * C, KP, device=dev) / C ** 0.5, 3, C)
pk1 = lib.pack_weight_planes(torch.randn(C, KP, device=dev) / C ** 0.5, 1, C)
pk0 = lib.pack_weight_planes(torch.randn(2 * C, 2 * KP, device=dev) / (2 * C) ** 0.5, 2, C)
pk2 = lib.pack_weight_planes(torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5, 1, C)
shapes = {
    "qkv f32+rot": lambda: lib.linear_planes(rows, C, 3, img, bnd, KP, pk3, lib.PL_F32, out=o32, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C),
    "merge+LN": lambda: lib.linear_planes(rows, C, 1, img, bnd, KP, pk1, lib.PL_LN, out_image=o_img, out_image_k=KP, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb),
    "mlp0": lambda: lib.linear_planes(rows, C, 2, img, bnd, KP, pk0, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=KP, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True),
    "mlp2+LN+res": lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=3 * C, out_image=o_img, out_image_k=KP, out_bound=o_b, gamma=g1, beta=b1, resid=xr, ldr=C, bound_resid=bnd, lnb=lnb)}
def t(f, n=15):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(30): shapes["mlp0"]()
for m16 in ("1", "0"):
    os.environ["DR_PG_M16"] = m16
    for name, f in shapes.items():
        out = []
        for tag, bits in (("full", 0), ("main loop only", 1), ("- fp32 stores", 4), ("- image stores", 8), ("- residual loads", 16), ("- tables", 32), ("- all memory", 60)):
            os.environ["DR_PG_NOEPI"] = str(bits)
            out.append("%s %6.1f" % (tag, min(t(f), t(f))))
        print("M16=%s %-12s " % (m16, name) + " | ".join(out))
