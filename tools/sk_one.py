import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(B, 256, 256, device="cuda:0") * 2
a = torch.tensor(1.0, device="cuda:0")
out = lib.sinkhorn(x, a, 3)
for _ in range(3): lib.sinkhorn(x, a, 3, out=out)
torch.cuda.synchronize()
