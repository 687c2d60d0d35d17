import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
from diffreg_hip import lib
torch.manual_seed(0)
dev = "cuda:0"
for (P, Lq, Lk, H, d) in ((1, 32, 32, 1, 108), (2, 128, 64, 4, 108), (3, 200, 256, 4, 108), (2, 96, 80, 4, 132), (2, 64, 160, 4, 64)):
    C = H * d
    q = torch.randn(P, Lq, C, device=dev); k = torch.randn(P, Lk, C, device=dev); v = torch.randn(P, Lk, C, device=dev) * 3
    o = lib.attention_planes(q, k, v, H)
    qh = q.double().view(P, Lq, H, d).transpose(1, 2); kh = k.double().view(P, Lk, H, d).transpose(1, 2); vh = v.double().view(P, Lk, H, d).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) / d ** 0.5, -1) @ vh).transpose(1, 2).reshape(P, Lq, C)
    err = (o.double() - ref).abs()
    print(P, Lq, Lk, H, d, "max err %.3e" % err.max().item(), "ref max %.2f" % ref.abs().max().item(), "nan", int(torch.isnan(o).sum()))
    if err.max() > 1e-3:
        e = err[0]
        bad_q = torch.nonzero(e.amax(1) > 1e-3).flatten()[:10].tolist(); bad_f = torch.nonzero(e.amax(0) > 1e-3).flatten()[:16].tolist()
        print("   bad queries", bad_q, "bad features", bad_f)
