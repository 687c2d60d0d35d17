"""cfg5 at ONE pair per call (the reference's actual mode, EXP/model.py:284): the f32-input MFMA kernels (the size rule's choice below 4 096 token rows)
against the plane-image path forced on (DR_LOOP_PLANES_FORCE), also at 2 pairs per call; interleaved rounds."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import _time_calls
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine2D3D
dev = torch.device("cuda:0")
N, M, steps, mc = 1024, 2048, 10, 200.0
Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
W = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in Wn.items()}
res = {}
for P in (1, 2):
    prs = [synth.make_pair_2d3d(N, M, 60 + i, weights=Wn) for i in range(P)]
    args = [torch.from_numpy(np.stack([p[k] for p in prs])).to(dev) for k in ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")]
    outs = {}
    for rnd in range(3):
        for planes in (False, True):
            eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=dev, planes=planes)
            kw = dict(zip(eng._ARGS, args))
            t = _time_calls(lambda: eng.run_static(slot=0, graph=True, **kw), warm=2, reps=5) * 1e3
            res.setdefault("P=%d planes=%s" % (P, planes), []).append(round(t, 2))
            outs[planes] = eng.run_static(slot=0, graph=True, **kw)["conf_matrix_pred"].clone()
            del eng
    res["P=%d max |conf(planes) - conf(f32)|" % P] = float((outs[True] - outs[False]).abs().max())
for k, v in res.items(): print(k, v)
print(json.dumps(res))
