"""DR_PG_STAMPS=1 python tools/pgemm_stamps.py : phase stamps (us) of workgroup 0 of the plane GEMM launches"""
import os, sys, ctypes
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
g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
lnb = lib.ln_bound(g1, b1)
Wm = torch.randn(C, C, device=dev) / C ** 0.5
pkm = lib.pack_weight_planes(Wm, 1, C)
W1 = torch.randn(2 * C, 2 * C, device=dev) / (2 * C) ** 0.5
pk1 = lib.pack_weight_planes(W1, 2, C)
W2 = torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5
pk2 = lib.pack_weight_planes(W2, 1, C)
o2 = torch.empty(rows, C, device=dev)
msg_img = torch.zeros_like(img); msg_b = torch.ones(rows, device=dev)
hid_img = torch.zeros(lib.raw().dr_plane_image_bytes(rows, 2 * C), dtype=torch.uint8, device=dev); hid_b = torch.ones(rows, device=dev)
o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev)

def stamps(name, f):
    for _ in range(5): f()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    lib.check(lib.raw().dr_debug_pgemm_stamps(buf))
    t = [b / 100.0 for b in buf]
    t0 = t[0]
    st = [t[8 + i] - t0 for i in range(0, 27)]
    print(name)
    print("  prologue wait %.2f | main loop %.2f | barrier %.2f | transpose %.2f | ln/f32 %.2f | end %.2f (us since start)" %
          (t[1] - t0, t[2] - t0, t[3] - t0, t[4] - t0, (t[5] if t[5] > t[4] else t[6]) - t0, t[6] - t0))
    for gname, o in (("wave 0 (group 0)", 64), ("wave 4 (group 1)", 80)):
        c = [buf[o + i] for i in range(7)]
        print("  stage 10 %s cycles: start->frags %d | burst1 %d | vmcnt %d | barrier %d | lgkm %d | burst2 %d | total %d" %
              (gname, c[6] - c[0], c[1] - c[6], c[2] - c[1], c[3] - c[2], c[4] - c[3], c[5] - c[4], c[5] - c[0]))
    print("  stage pairs (us):", " ".join("%.2f" % (st[i] - (st[i - 1] if i else t[1] - t0)) for i in range(14)))

stamps("F32 432x432", lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pkm, lib.PL_F32, out=o2, ldo=C))
stamps("merge+LN planes", lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pkm, lib.PL_LN, out_image=msg_img, out_image_k=C, out_bound=msg_b, gamma=g1, beta=b1, lnb=lnb))
stamps("mlp0", lambda: lib.linear_planes(rows, C, 2, img, bnd, C, pk1, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=hid_img, out_image_k=2 * C, out_bound=hid_b, relu=True))
stamps("mlp2+LN+res", lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o2, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb))
