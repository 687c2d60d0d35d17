"""Phase stamps (us) of the LAST Procrustes launch of a B = 1 engine run (needs DR_DIAGNOSTICS=1): load, K, level 1, select, take, reduce, SVD"""
import os, sys, ctypes
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import torch
from diffreg_hip import synth, lib
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import weights, pair
variant, N, M, steps, mc = "3dmatch", 256, 256, 20, 200
v = synth.VARIANTS[variant]
eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                    sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device="cuda:0")
_, p = pair(variant, N, M, 13)
a = [p[k].to("cuda:0") for k in ("f_s", "f_t", "p_s", "p_t", "x_T")]
for _ in range(2): eng.run(*a, graph=False)
torch.cuda.synchronize()
st = (ctypes.c_longlong * 8)()
lib.check(lib.raw().dr_debug_procrustes_stamps(st))
print([round((st[i + 1] - st[i]) / 100.0, 1) for i in range(7)])
