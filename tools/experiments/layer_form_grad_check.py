"""_GeometryAttentionLayerG (sinusoidal / no-code forms) against torch autograd through the reference's arithmetic in float64 (one layer, random weights)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
from diffreg_hip.autograd import _GeometryAttentionLayerG
torch.manual_seed(0)
B, L, S, C, H = 1, 96, 80, 432, 4
d = C // H
dev = "cuda:0"
x, y = torch.randn(B, L, C, dtype=torch.float64) * 0.5, torch.randn(B, S, C, dtype=torch.float64) * 0.5
px, py = torch.randn(B, L, C, dtype=torch.float64), torch.randn(B, S, C, dtype=torch.float64)
W = [torch.randn(C, C, dtype=torch.float64) / C ** 0.5 for _ in range(4)] + [torch.randn(2 * C, 2 * C, dtype=torch.float64) / (2 * C) ** 0.5,
     torch.randn(C, 2 * C, dtype=torch.float64) / (2 * C) ** 0.5] + [torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.1,
     torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.1]
Rw = torch.randn(B, L, C, dtype=torch.float64)
for form in ("sin", "none"):
    for self_att in (False, True):
        xr = x.clone().requires_grad_(True)
        yr = xr if self_att else y.clone().requires_grad_(True)
        pyy = px if self_att else py
        Wr = [w.clone().requires_grad_(True) for w in W]
        Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2 = Wr
        q = xr + px if form == "sin" else xr
        k = yr + pyy if form == "sin" else yr
        qw, kw, vw = (q @ Wq.t()).view(B, -1, H, d), (k @ Wk.t()).view(B, -1, H, d), (yr @ Wv.t()).view(B, -1, H, d)
        a = torch.einsum("nlhd,nshd->nlsh", qw, kw) / d ** 0.5
        a = torch.softmax(a, 2)
        o = torch.einsum("nlsh,nshd->nlhd", a, vw).reshape(B, -1, C)
        m = torch.nn.functional.layer_norm(o @ Wm.t(), (C,), g1, b1)
        f = torch.nn.functional.layer_norm(torch.relu(torch.cat([xr, m], 2) @ W0.t()) @ W2.t(), (C,), g2, b2)
        e = xr + f
        (e * Rw).sum().backward()
        xd = x.float().to(dev).requires_grad_(True)
        yd = xd if self_att else y.float().to(dev).requires_grad_(True)
        Wd = [w.float().to(dev).requires_grad_(True) for w in W]
        pa, pb = (px.float().to(dev), pyy.float().to(dev)) if form == "sin" else (None, None)
        ed = _GeometryAttentionLayerG.apply(xd, yd, pa, pb, None, None, None, None, None, None, H, *Wd)
        (ed * Rw.float().to(dev)).sum().backward()
        rel = lambda g_, r_: float((g_.double().cpu() - r_).abs().max() / r_.abs().max())
        print(form, "self" if self_att else "cross", "out %.2e" % rel(ed.detach(), e.detach()), "gx %.2e" % rel(xd.grad, xr.grad),
              "" if self_att else "gy %.2e" % rel(yd.grad, yr.grad), " ".join("%.1e" % rel(a_.grad, b_.grad) for a_, b_ in zip(Wd, Wr)))
