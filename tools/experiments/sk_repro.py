import os, sys, torch
sys.path.insert(0, "diff-reg_amd"); sys.path.insert(0, ".")
import numpy as np
from diffreg_hip import lib, synth
lib.ensure_init(); lib.raw().dr_debug_enable_env(1)
os.environ["DR_SK_XCD"] = os.environ.get("XCD", "0"); os.environ["DR_SK_ZERO_MEMSET"] = os.environ.get("ZM", "1")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
B, N, M, nv, mv = [int(v) for v in os.environ.get("SHAPE", "5,1000,1530,1000,1530").split(",")]
raw = torch.cat([T(3.0 * synth.hash_normal(17 + b, N * 1000 + M, (1, N, M))) for b in range(B)]).double()
sm = (torch.arange(N)[None].expand(B, N) < torch.tensor([nv - 3 * b for b in range(B)])[:, None]).cuda()
tm = (torch.arange(M)[None].expand(B, M) < torch.tensor([mv - 5 * b for b in range(B)])[:, None]).cuda()
x = raw.cuda(); a = torch.tensor(1.0).cuda()
print("inputs ready", flush=True)
for mode in os.environ.get("MODES", "masks,nomasks").split(","):
    for i in range(4):
        if mode == "masks": o = lib.sinkhorn(x, a, 3, sm, tm, apply_mask=True, out_f32=True)
        else: o = lib.sinkhorn(x, a, 3, out_f32=True)
        torch.cuda.synchronize()
        print(mode, "call", i, "ok; checksum", float(o.double().sum()), "nan", int(torch.isnan(o).sum()), flush=True)
