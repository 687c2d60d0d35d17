"""time per launch of the four layer GEMM shapes of the plane path at ROWS rows (HIP events; DR_DIAGNOSTICS=1 DR_PG_HALF=0 selects the 128-row geometry)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
C = 432
for rows in [int(r) for r in os.environ.get("ROWS", "32768,65536").split(",")]:
    x = torch.randn(rows, C, device=dev)
    img, bnd = lib.planes_from_f32(x)
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    lnb = lib.ln_bound(g1, b1)
    msg_img, msg_b = lib.planes_from_f32(torch.randn(rows, C, device=dev))
    hid_img, hid_b = lib.planes_from_f32(torch.randn(rows, 2 * C, device=dev))
    o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, 3 * C, device=dev)
    h_img = torch.zeros_like(hid_img); h_b = torch.zeros(rows, device=dev)
    ang = torch.rand(rows, C // 2, device=dev); cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    pk3 = lib.pack_weight_planes(torch.randn(3 * C, C, device=dev) / C ** 0.5, 3, C)
    pk1 = lib.pack_weight_planes(torch.randn(C, C, device=dev) / C ** 0.5, 1, C)
    pk0 = lib.pack_weight_planes(torch.randn(2 * C, 2 * C, device=dev) / (2 * C) ** 0.5, 2, C)
    pk2 = lib.pack_weight_planes(torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5, 1, C)
    shapes = {
        "qkv f32+rot": (lambda: lib.linear_planes(rows, C, 3, img, bnd, C, pk3, lib.PL_F32, out=o32, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C), 3 * C * C),
        "merge+LN": (lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pk1, lib.PL_LN, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb), C * C),
        "mlp0": (lambda: lib.linear_planes(rows, C, 2, img, bnd, C, pk0, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True), 4 * C * C),
        "mlp2+LN+res": (lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=3 * C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb), 2 * C * C)}
    for name, (f, kn) in shapes.items():
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("rows %6d %-12s %7.1f us  %6.1f TFLOP/s" % (rows, name, us, 2.0 * rows * kn / us / 1e6))
