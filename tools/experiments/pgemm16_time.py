"""the four layer GEMM shapes at 65 536 rows: the 32x32x16 main loop (DR_PG_M16=0) against the 16x16x32 one (DR_PG_M16=1, the default); alternating, HIP events.
KP=448 pads K to whole chunk pairs (no virtual chunk)."""
import os, sys, torch
os.environ["DR_DIAGNOSTICS"] = "1"; os.environ["DR_PG_HALF"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
C, KP = 432, int(os.environ.get("KP", "432"))
for rows in [int(r) for r in os.environ.get("ROWS", "65536").split(",")]:
    x = torch.randn(rows, KP, device=dev); x[:, C:] = 0
    img, bnd = lib.planes_from_f32(x)
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    lnb = lib.ln_bound(g1, b1)
    msg_img, msg_b = lib.planes_from_f32(torch.randn(rows, KP, device=dev))
    hid_img, hid_b = lib.planes_from_f32(torch.randn(rows, 2 * C, device=dev))
    o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, 3 * C, device=dev)
    h_img = torch.zeros_like(hid_img); h_b = torch.zeros(rows, device=dev)
    xr = torch.randn(rows, C, device=dev)
    ang = torch.rand(rows, C // 2, device=dev); cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    pk3 = lib.pack_weight_planes(torch.randn(3 * C, KP, device=dev) / C ** 0.5, 3, C)
    pk1 = lib.pack_weight_planes(torch.randn(C, KP, device=dev) / C ** 0.5, 1, C)
    pk0 = lib.pack_weight_planes(torch.randn(2 * C, 2 * KP, device=dev) / (2 * C) ** 0.5, 2, C)
    pk2 = lib.pack_weight_planes(torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5, 1, C)
    shapes = {
        "qkv f32+rot": (lambda: lib.linear_planes(rows, C, 3, img, bnd, KP, pk3, lib.PL_F32, out=o32, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C), 3 * C * KP),
        "qkv f32 norot": (lambda: lib.linear_planes(rows, C, 3, img, bnd, KP, pk3, lib.PL_F32, out=o32, ldo=3 * C, blk_stride=C), 3 * C * KP),
        "qkv 1blk+rot": (lambda: lib.linear_planes(rows, C, 1, img, bnd, KP, pk3, lib.PL_F32, out=o32, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C), C * KP),
        "merge+LN": (lambda: lib.linear_planes(rows, C, 1, img, bnd, KP, pk1, lib.PL_LN, out_image=o_img, out_image_k=KP, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb), C * KP),
        "mlp0": (lambda: lib.linear_planes(rows, C, 2, img, bnd, KP, pk0, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=KP, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True), 4 * C * KP),
        "mlp2+LN+res": (lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=3 * C, out_image=o_img, out_image_k=KP, out_bound=o_b, gamma=g1, beta=b1, resid=xr, ldr=C, bound_resid=bnd, lnb=lnb), 2 * C * C)}
    def t(f, n=20):
        for _ in range(4): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for _ in range(30): shapes["mlp0"][0]()          # warm the clocks
    for name, (f, kn) in shapes.items():
        res = {"0": [], "1": []}
        for rnd in range(3):
            for m in ("0", "1"):
                os.environ["DR_PG_M16"] = m
                res[m].append(t(f))
        a, b = min(res["0"]), min(res["1"])
        print("rows %6d %-12s 32x32x16 %7.1f us (%5.1f TF)   16x16x32 %7.1f us (%5.1f TF)   %+.1f %%" % (rows, name, a, 2.0 * rows * kn / a / 1e6, b, 2.0 * rows * kn / b / 1e6, (a / b - 1) * 100), res)
