"""Interleaved A/B of one debug knob inside the loops: cfg5 (8 pairs per call; one call and two concurrent calls) and cfg2 (128 pairs x 2 streams).
KNOB=DR_ATTN_XCD VALUES=1,0 python tools/experiments/knob_ab_time.py      (the library reads DR_* only behind the debug gate, opened here)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import _time_calls, HEAD_GAIN
from diffreg_hip import lib, synth
from diffreg_hip.engine import DenoiseEngine, DenoiseEngine2D3D
lib.ensure_init(); lib.raw().dr_debug_enable_env(1)
dev = torch.device("cuda:0")
knob, values = os.environ["KNOB"], os.environ.get("VALUES", "1,0").split(",")
res = {}

def cfg5(v):
    N, M, steps, mc, P = 1024, 2048, 10, 200.0, 8
    Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
    W = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in Wn.items()}
    eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=dev)
    prs = [synth.make_pair_2d3d(N, M, 60 + i % 4, weights=Wn) for i in range(P)]
    args = [torch.from_numpy(np.stack([p[k] for p in prs])).to(dev) for k in ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")]
    kw = dict(zip(eng._ARGS, args))
    one = _time_calls(lambda: eng.run_static(slot=0, graph=True, **kw), warm=2, reps=4) * 1e3
    two = _time_calls(lambda: eng.run_streams([kw, kw], 2), warm=2, reps=4) * 1e3
    return {"cfg5 one 8-pair call ms": round(one, 2), "cfg5 two concurrent calls ms": round(two, 2)}

def cfg2(v):
    variant, N, M, steps, mc, P = "3dmatch", 256, 256, 20, 200.0, 128
    vv = synth.VARIANTS[variant]
    W = {k: torch.from_numpy(a) for k, a in synth.make_weights(vv["C"], seed=7, head_gain=HEAD_GAIN).items()}
    eng = DenoiseEngine(W, variant=variant, C=vv["C"], H=vv["H"], voxel=vv["voxel"], origin=vv["origin"], steps=steps, sk_iters=vv["skh_iters"],
                        sample_rate=vv["sample_rate"], max_condition_num=mc, n_layers=vv["n_layers"], device=dev)
    def group(seed0):
        prs = [synth.make_pair(N, M, vv["C"], seed=seed0 + i % 8) for i in range(P)]
        st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev)
        return dict(src_feats=st("src_feats"), tgt_feats=st("tgt_feats"), s_pcd=st("s_pcd"), t_pcd=st("t_pcd"), x_T=st("x_T"))
    g0, g1 = group(100), group(200)
    two = _time_calls(lambda: eng.run_streams([g0, g1], 2), warm=2, reps=3) * 1e3
    return {"cfg2 256 pairs in two streams ms": round(two, 2)}

which = os.environ.get("WHICH", "cfg5,cfg2").split(",")
for rnd in range(3):
    for v in values:
        os.environ[knob] = v
        r = {}
        if "cfg5" in which: r.update(cfg5(v))
        if "cfg2" in which: r.update(cfg2(v))
        for k, x in r.items():
            res.setdefault("%s = %s: %s" % (knob, v, k), []).append(x)
for k, x in sorted(res.items()): print(k, x)
print(json.dumps(res))
