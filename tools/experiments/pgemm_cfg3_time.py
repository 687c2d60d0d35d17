"""the four layer-GEMM shapes of cfg3 (C = 528, 4 heads of d = 132 padded to 144: q | k | v blocks of 576 columns) at ROWS rows (8 192 = 8 pairs x (512 + 512)):
per-launch time with the whole kernel and with the main loop alone (DR_PG_NOEPI=1), HIP events.  KNOB / KVALS: an A/B knob of the build under test."""
import os, sys, torch
os.environ["DR_DIAGNOSTICS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
C, H, d, dp = 528, 4, 132, 144
Cq = H * dp
KNOB, KVALS = os.environ.get("KNOB", ""), os.environ.get("KVALS", "0,1").split(",")
WIDE = os.environ.get("WIDE", "0") == "1"        # q | k | v and mlp0 on the wide-wave kernel (128 x 288 workgroups)
for rows in [int(r) for r in os.environ.get("ROWS", "8192").split(",")]:
    x = torch.randn(rows, C, device=dev)
    img, bnd = lib.planes_from_f32(x)
    att_img, att_b = lib.planes_from_f32(torch.randn(rows, Cq, device=dev))
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    lnb = lib.ln_bound(g1, b1)
    msg_img, msg_b = lib.planes_from_f32(torch.randn(rows, C, device=dev))
    o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, C, device=dev)
    q_img = torch.zeros(3, att_img.numel(), dtype=torch.uint8, device=dev); q_b = torch.zeros(3, rows, device=dev)
    h_img = torch.zeros(lib._lib.dr_plane_image_bytes(rows, 2 * C), dtype=torch.uint8, device=dev); h_b = torch.zeros(rows, device=dev)
    xr = torch.randn(rows, C, device=dev)
    ang = torch.rand(rows, C // 2, device=dev); cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    pk3 = lib.pack_weight_planes(torch.randn(3 * Cq, C, device=dev) / C ** 0.5, 3, Cq, wide=WIDE)
    pk1 = lib.pack_weight_planes(torch.randn(C, Cq, device=dev) / C ** 0.5, 1, C)
    pk0 = lib.pack_weight_planes(torch.randn(2 * C, 2 * C, device=dev) / (2 * C) ** 0.5, 2, C, wide=WIDE)
    pk2 = lib.pack_weight_planes(torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5, 1, C)
    o3 = torch.empty(rows, 3 * Cq, device=dev)
    shapes = {
        "qkv f32+rot": (lambda: lib.linear_planes(rows, Cq, 3, img, bnd, C, pk3, lib.PL_F32, out=o3, ldo=3 * Cq, blk_stride=Cq, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C, wide=WIDE), 3 * C * C),
        "merge+LN": (lambda: lib.linear_planes(rows, C, 1, att_img, att_b, Cq, pk1, lib.PL_LN, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb), C * C),
        "mlp0": (lambda: lib.linear_planes(rows, C, 2, img, bnd, C, pk0, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True, wide=WIDE), 4 * C * C),
        "mlp2+LN+res": (lambda: lib.linear_planes(rows, C, 1, h_img, h_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=xr, ldr=C, bound_resid=bnd, lnb=lnb), 2 * C * C)}
    def t(f, n=20):
        for _ in range(4): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for _ in range(60): shapes["mlp0"][0]()          # warm the clocks
    tot = {}
    for name, (f, kn) in shapes.items():
        for kv in (KVALS if KNOB else [""]):
            if KNOB: os.environ[KNOB] = kv
            res = {}
            for noepi in ("0", "1"):
                os.environ["DR_PG_NOEPI"] = noepi
                res[noepi] = min(t(f) for _ in range(3))
            os.environ["DR_PG_NOEPI"] = "0"
            tot[kv] = tot.get(kv, 0.0) + res["0"]
            print("rows %6d %-12s %s whole %7.1f us (%5.1f TF, frac %.3f)   main loop alone %7.1f us (%5.1f TF)" % (
                rows, name, ("%s=%s" % (KNOB, kv)) if KNOB else "", res["0"], 2.0 * rows * kn / res["0"] / 1e6, 2.0 * rows * kn / res["0"] / 1e6 / 838.9,
                res["1"], 2.0 * rows * kn / res["1"] / 1e6))
    print("rows %d: sum of the four launches" % rows, {k: round(v, 1) for k, v in tot.items()})
