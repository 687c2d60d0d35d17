import sys; sys.path.insert(0, "/root/repo/diff-reg_amd")
import torch
from diffreg_hip import lib
torch.manual_seed(0)
for (nb, R, N, K) in ((4, 256, 256, 108), (4, 256, 108, 256), (4, 64, 48, 108), (8, 96, 80, 108), (1, 256, 256, 432), (4, 108, 256, 256), (4, 256, 432, 256)):
    a = torch.randn(nb, R, K, device="cuda"); b = torch.randn(nb, N, K, device="cuda")
    o = lib.bmm_nt(a, b)
    ref = torch.stack([lib.linear(a[i], b[i]) for i in range(nb)])
    r64 = a.double() @ b.double().transpose(1, 2)
    print((nb, R, N, K), "bmm vs loop %.3e" % (o - ref).abs().max().item(), "bmm vs f64 %.3e" % (o - r64).abs().max().item(), "loop vs f64 %.3e" % (ref - r64).abs().max().item())
