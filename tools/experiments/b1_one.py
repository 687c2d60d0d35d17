"""One pair (B = 1, N = M = 256, 20 steps) through the engine: a few graph replays -- the target of rocprofv3 --kernel-trace --stats"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import weights, pair
DEV = "cuda:0"
variant, N, M, steps, mc = "3dmatch", int(os.environ.get("B1_N", "256")), int(os.environ.get("B1_M", "256")), 20, 200
v = synth.VARIANTS[variant]
planes = {"1": True, "0": False}.get(os.environ.get("B1_PLANES", ""), None)
eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                    sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, planes=planes)
_, p = pair(variant, N, M, 13)
a = [p[k].to(DEV) for k in ("f_s", "f_t", "p_s", "p_t", "x_T")]
graph = os.environ.get("B1_GRAPH", "1") == "1"
for _ in range(3): eng.run(*a, graph=graph)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): eng.run(*a, graph=graph)
torch.cuda.synchronize()
print("ms per pair %.3f (graph=%s, planes=%s)" % ((time.perf_counter() - t0) / n * 1e3, graph, planes))
