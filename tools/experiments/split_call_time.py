"""One call's batch split over concurrent streams INSIDE the call: cfg5's 8 pairs as 2 x 4 / 4 x 2, cfg3's 8 pairs as 2 x 4 (every group a
captured graph on its own stream).  Question: do two half-size groups beat one 8-pair call (whose launches run in lockstep: load phase, MFMA
phase, store phase on all CUs at once)?  python tools/experiments/split_call_time.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import _time_calls, HEAD_GAIN
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine, DenoiseEngine2D3D
dev = torch.device("cuda:0")
out = []

def cfg5():
    N, M, steps, mc = 1024, 2048, 10, 200.0
    Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
    W = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in Wn.items()}
    eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=dev)
    distinct = [synth.make_pair_2d3d(N, M, 60 + i, weights=Wn) for i in range(4)]
    def kw(P, off=0):
        prs = [distinct[(off + i) % 4] for i in range(P)]
        args = [torch.from_numpy(np.stack([p[k] for p in prs])).to(dev) for k in ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")]
        return dict(zip(eng._ARGS, args))
    k8 = kw(8)
    t = _time_calls(lambda: eng.run_static(slot=0, graph=True, **k8), warm=3, reps=4)
    out.append({"cfg": "cfg5", "form": "1 x 8", "ms": t * 1e3, "pairs_per_s": 8 / t})
    for g, p in ((2, 4), (4, 2)):
        groups = [kw(p, i * p) for i in range(g)]
        t = _time_calls(lambda: eng.run_streams(groups, g), warm=3, reps=4)
        out.append({"cfg": "cfg5", "form": "%d x %d" % (g, p), "ms": t * 1e3, "pairs_per_s": 8 / t})

def cfg3():
    variant, N, M, steps, mc = "4dmatch", 512, 512, 20, 40.0
    v = synth.VARIANTS[variant]
    W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=HEAD_GAIN).items()}
    eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                        sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=dev)
    def group(seed0, P):
        prs = [synth.make_pair(N, M, v["C"], seed=seed0 + i) for i in range(P)]
        st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev)
        noise = torch.from_numpy(np.stack([synth.step_noise(N, M, seed0 + i, steps) for i in range(P)], 1)).to(dev)
        return dict(src_feats=st("src_feats"), tgt_feats=st("tgt_feats"), s_pcd=st("s_pcd"), t_pcd=st("t_pcd"), x_T=st("x_T"),
                    src_mask=torch.ones(P, N, dtype=torch.bool, device=dev), tgt_mask=torch.ones(P, M, dtype=torch.bool, device=dev), noise=noise)
    g8 = group(300, 8)
    t = _time_calls(lambda: eng.run(graph=True, borrow=True, **g8), warm=3, reps=4)
    out.append({"cfg": "cfg3", "form": "1 x 8", "ms": t * 1e3, "pairs_per_s": 8 / t})
    for g, p in ((2, 4), (4, 2)):
        groups = [group(300 + i * p, p) for i in range(g)]
        t = _time_calls(lambda: eng.run_streams(groups, g), warm=3, reps=4)
        out.append({"cfg": "cfg3", "form": "%d x %d" % (g, p), "ms": t * 1e3, "pairs_per_s": 8 / t})

cfg5(); cfg3()
for o in out: print(o)
print(json.dumps(out))
