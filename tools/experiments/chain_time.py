"""cfg5 at P pairs per call: three launches per layer call (DR_PG_CHAIN=0) against the chained launch for every call (DR_PG_CHAIN_MIN=0) and for the
calls the product chains (default rule); modes interleaved over three rounds in one process.  Needs DR_DIAGNOSTICS=1."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "diff-reg_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import _time_calls
from diffreg_hip import lib, synth
from diffreg_hip.engine import DenoiseEngine2D3D
dev = torch.device("cuda:0")
N, M, steps, mc = 1024, 2048, 10, 200.0
Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
W = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in Wn.items()}
distinct = [synth.make_pair_2d3d(N, M, 60 + i, weights=Wn) for i in range(4)]
MODES = {"three launches": {"DR_PG_CHAIN": "0"}, "chain everywhere": {"DR_PG_CHAIN_MIN": "0"}, "chain by the rule": {}}
out = []
for P in [int(a) for a in os.environ.get("PS", "8,4").split(",")]:
    prs = [distinct[i % 4] for i in range(P)]
    args = [torch.from_numpy(np.stack([p[k] for p in prs])).to(dev) for k in ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")]
    res = {m: [] for m in MODES}
    launches = {}
    for rnd in range(3):
        for m, env in MODES.items():
            for k in ("DR_PG_CHAIN", "DR_PG_CHAIN_MIN"):
                os.environ.pop(k, None)
            os.environ.update(env)
            eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=dev)
            kw = dict(zip(eng._ARGS, args))
            res[m].append(_time_calls(lambda: eng.run_static(slot=0, graph=True, **kw), warm=2, reps=4) * 1e3)
            if rnd == 0:
                lib.prof_enable(True); eng.run(*args); prof = lib.prof_collect(); lib.prof_enable(False)
                launches[m] = prof["gemm_split"][0]
            del eng
    for m in MODES:
        out.append({"P": P, "mode": m, "ms_per_call": [round(x, 2) for x in res[m]], "median": float(np.median(res[m])), "pgemm_launches": launches[m]})
        print(out[-1])
print(json.dumps(out))
