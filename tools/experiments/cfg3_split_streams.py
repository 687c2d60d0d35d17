"""cfg3 (4DMatch 512^2, C = 528, 20 steps, 8 pairs): ONE 8-pair call on one stream against the same 8 pairs as pair groups run concurrently
(run_streams: one captured graph per group).  Prints ms per 8 pairs and pairs/s for every arrangement."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
variant, N, M, steps, mc, P = "4dmatch", 512, 512, 20, 40.0, 8
v = synth.VARIANTS[variant]
W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=24.0).items()}
eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                    sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device="cuda:0")
prs = [synth.make_pair(N, M, v["C"], seed=300 + i) for i in range(P)]
def group(idx):
    st = lambda k: torch.from_numpy(np.stack([prs[i][k] for i in idx])).cuda()
    noise = torch.from_numpy(np.stack([synth.step_noise(N, M, 300 + i, steps) for i in idx], 1)).cuda()
    n = len(idx)
    return dict(src_feats=st("src_feats"), tgt_feats=st("tgt_feats"), s_pcd=st("s_pcd"), t_pcd=st("t_pcd"), x_T=st("x_T"),
                src_mask=torch.ones(n, N, dtype=torch.bool, device="cuda"), tgt_mask=torch.ones(n, M, dtype=torch.bool, device="cuda"), noise=noise)
def timed(fn, warm=3, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
res = {}
g8 = group(range(8))
one = eng.run(graph=True, **g8)
ref_conf = one["conf_matrix_pred"].clone()
res["one 8-pair call, one stream"] = timed(lambda: eng.run(graph=True, **g8))
for ng, ns in ((2, 2), (4, 2), (4, 4), (8, 4)):
    sz = 8 // ng
    groups = [group(range(i * sz, (i + 1) * sz)) for i in range(ng)]
    outs = eng.run_streams(groups, ns)
    torch.cuda.synchronize()
    conf = torch.cat([o["conf_matrix_pred"] for o in outs])
    dev = (conf - ref_conf).abs().max().item()
    res["%d groups of %d pairs on %d streams" % (ng, sz, ns)] = timed(lambda: eng.run_streams(groups, ns))
    res["%d groups of %d: max |conf - conf of the 8-pair call|" % (ng, sz)] = dev
out = {k: ({"ms_per_8_pairs": round(t * 1e3, 2), "pairs_per_s": round(8 / t, 1)} if "max" not in k else t) for k, t in res.items()}
print(json.dumps(out, indent=1))
