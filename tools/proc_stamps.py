import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
from diffreg_hip import lib
N = M = 256
g = torch.Generator().manual_seed(0)
x = torch.randn(1, N, M, generator=g) * 3
x[0, torch.arange(N), torch.randperm(N, generator=g)] += float(os.environ.get("PEAK", "8"))
conf = lib.sinkhorn(x.cuda(), torch.tensor(1.0).cuda(), 3)
ps, pt = torch.rand(1, N, 3).cuda(), torch.rand(1, M, 3).cuda()
for rep in range(3):
    lib.procrustes(conf, ps, pt, None, None, 1.0, 200.0)
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 8)()
    lib.check(lib.raw().dr_debug_procrustes_stamps(st))
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(7)]
    print("us: load %.1f | K/setup %.1f | level1 %.1f | level2-select %.1f | take %.1f | reduce %.1f | svd %.1f | tail %.1f | total %.1f" % (
        d[0], 0, d[1], d[2], d[3], d[4], d[5], d[6], (st[7] - st[0]) / 100.0))
