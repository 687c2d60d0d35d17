import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from diffreg_hip import synth, lib
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import T, weights, pair
DEV = "cuda:0"
def run(variant, N, M, P, planes=None, masked=True):
    v = synth.VARIANTS[variant]
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=1,
                        sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=40, n_layers=v["n_layers"], device=DEV, planes=planes)
    prs = [pair(variant, N, M, 60 + i)[1] for i in range(P)]
    cat = lambda k: torch.cat([q[k] for q in prs]).to(DEV)
    ms = torch.stack([torch.arange(N) < (N - 7 * i if masked else N) for i in range(P)]).to(DEV); mt = torch.stack([torch.arange(M) < (M - 11 * i if masked else M) for i in range(P)]).to(DEV)
    singles = []
    for i, q in enumerate(prs):
        so, to, conf = eng.denoise_match(q["f_s"].to(DEV), q["f_t"].to(DEV), q["p_s"].to(DEV), q["p_t"].to(DEV), ms[i:i + 1], mt[i:i + 1])
        singles.append(so.clone())
    outs = []
    for rep in range(2):
        so, to, conf = eng.denoise_match(cat("f_s"), cat("f_t"), cat("p_s"), cat("p_t"), ms, mt)
        outs.append(so.clone())
    ds = max((outs[0][i] - singles[i][0]).abs().max().item() for i in range(P))
    print(variant, N, M, "P", P, "planes", planes, "masked", masked, "| batch vs single %.2e | run-to-run %.2e" % (ds, (outs[0] - outs[1]).abs().max().item()))
run("4dmatch", 512, 512, 8)
run("4dmatch", 512, 512, 8, masked=False)
run("4dmatch", 256, 256, 16)
run("4dmatch", 128, 128, 64)
run("3dmatch", 512, 512, 8, planes=False)
run("3dmatch", 256, 256, 32, planes=False)
run("3dmatch", 512, 512, 8, planes=True)
