"""One shape of the plane-image GEMM, a few launches: the target of rocprofv3 --pmc / --kernel-trace runs.
   MODE = f32 | ln | mlp0 | mlp2   ROWS (default 32768)   N launches (default 10)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
dev = torch.device("cuda:0")
rows, C = int(os.environ.get("ROWS", "32768")), 432
mode = os.environ.get("MODE", "mlp0")
x = torch.randn(rows, C, device=dev)
img, bnd = lib.planes_from_f32(x)
g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
lnb = lib.ln_bound(g1, b1)
msg_img, msg_b = lib.planes_from_f32(torch.randn(rows, C, device=dev))
hid_img, hid_b = lib.planes_from_f32(torch.randn(rows, 2 * C, device=dev))
o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, C, device=dev)
h_img = torch.zeros_like(hid_img); h_b = torch.zeros(rows, device=dev)
if mode == "f32":
    pk = lib.pack_weight_planes(torch.randn(C, C, device=dev) / C ** 0.5, 1, C)
    f = lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pk, lib.PL_F32, out=o32, ldo=C)
elif mode == "ln":
    pk = lib.pack_weight_planes(torch.randn(C, C, device=dev) / C ** 0.5, 1, C)
    f = lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pk, lib.PL_LN, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb)
elif mode == "mlp0":
    pk = lib.pack_weight_planes(torch.randn(2 * C, 2 * C, device=dev) / (2 * C) ** 0.5, 2, C)
    f = lambda: lib.linear_planes(rows, C, 2, img, bnd, C, pk, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True)
else:
    pk = lib.pack_weight_planes(torch.randn(C, 2 * C, device=dev) / (2 * C) ** 0.5, 1, C)
    f = lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb)
torch.cuda.synchronize()
for _ in range(int(os.environ.get("N", "10"))):
    f()
torch.cuda.synchronize()
