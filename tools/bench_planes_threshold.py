"""loop time per pair vs batch size with the plane path forced on / off (where does it start to pay?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import torch
sys.argv = [sys.argv[0]]
import bench
dev = torch.device("cuda:0")
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
variant = os.environ.get("VARIANT", "3dmatch")
v = synth.VARIANTS[variant]
N = M = int(os.environ.get("N", "256"))
W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=24.0).items()}
for P in (1, 2, 4, 8, 16, 32):
    _, inp = bench.make_inputs(variant, P, N, M, 100, dev)
    res = {}
    for planes in (False, True):
        eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=20, sk_iters=v["skh_iters"],
                            sample_rate=v["sample_rate"], max_condition_num=200.0 if variant == "3dmatch" else 40.0, n_layers=v["n_layers"], device=dev, planes=planes)
        kw = {}
        if variant == "4dmatch":
            kw = dict(src_mask=torch.ones(P, N, dtype=torch.bool, device=dev), tgt_mask=torch.ones(P, M, dtype=torch.bool, device=dev),
                      noise=torch.randn(20, P, N, M, device=dev))
        f = lambda: eng.run(inp["f_s"], inp["f_t"], inp["p_s"], inp["p_t"], inp["x_T"], graph=True, **kw)
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        res[planes] = (time.perf_counter() - t0) / 3 * 1e3
    print("%s N=%d P=%d rows=%d: f32 kernels %.2f ms/call, plane path %.2f ms/call" % (variant, N, P, P * 2 * N, res[False], res[True]))
