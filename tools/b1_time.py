"""Single-pair latency (B = 1, 20 steps, graph replay) at N = M = 256 and at a real 3DMatch pair's coarse size (564 x 629), and the launch
count of one eager pass with its per-family times (HIP events)."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import bench
from diffreg_hip import lib, synth
res = {}
for name, (N, M) in (("n256", (256, 256)), ("real_564x629", (564, 629))):
    W, eng = bench.make_engine("3dmatch", 20, 200.0, "cuda:0")
    p = synth.make_pair(N, M, 432, seed=9000)
    a = [torch.from_numpy(p[k])[None].to("cuda:0") for k in ("src_feats", "tgt_feats", "s_pcd", "t_pcd", "x_T")]
    dt = bench._time_calls(lambda: eng.run(*a, graph=True, borrow=True), warm=3, reps=20)
    lib.prof_enable(True)
    eng.run(*a, graph=False, borrow=True)
    prof = lib.prof_collect()
    lib.prof_enable(False)
    res[name] = dict(ms_per_pair=dt * 1e3, launches=sum(v[0] for v in prof.values()),
                     families={k: dict(n=v[0], ms=round(v[1], 3), avg_us=round(v[1] / v[0] * 1e3, 2)) for k, v in prof.items() if v[0]})
    print(name, json.dumps(res[name]), flush=True)
