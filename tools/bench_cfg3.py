"""BASELINE configs[2] shape (4DMatch: N = M = 512, C = 528, 20 steps, 8 pairs per call): seconds per call on the GPU
(a secondary line; bench.py is configs[1])."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
variant, N, M, steps, mc = "4dmatch", 512, 512, 20, 40.0
P = int(os.environ.get("P", "8"))
v = synth.VARIANTS[variant]
W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=24.0).items()}
eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                    sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device="cuda:0")
prs = [synth.make_pair(N, M, v["C"], seed=300 + i) for i in range(P)]
st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).cuda()
noise = torch.from_numpy(np.stack([synth.step_noise(N, M, 300 + i, steps) for i in range(P)], 1)).cuda()
ms = torch.ones(P, N, dtype=torch.bool, device="cuda"); mt = torch.ones(P, M, dtype=torch.bool, device="cuda")
run = lambda: eng.run(st_fs, st_ft, st_ps, st_pt, st_x, ms, mt, noise=noise, graph=os.environ.get("GRAPH", "1") == "1")
st_fs, st_ft, st_ps, st_pt, st_x = st("src_feats"), st("tgt_feats"), st("s_pcd"), st("t_pcd"), st("x_T")
for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
res = {"workload": "4DMatch N=M=512, C=528 (d_head 132), %d steps, %d pairs per call" % (steps, P), "gpu_s_per_call": dt, "gpu_pairs_per_s": P / dt,
       "gemm_path": "plane images, 576-column geometry (pgemm_kernel<9,3,...>); attention on plane images (fp16 hi/lo operands, 3 MFMA products)"}
# ---- per-family GPU time of one eager call (HIP events on the stream) and the roofline of the dominant family
from diffreg_hip import lib
eng.run(st_fs, st_ft, st_ps, st_pt, st_x, ms, mt, noise=noise, graph=False)
torch.cuda.synchronize()
lib.prof_enable(True)
eng.run(st_fs, st_ft, st_ps, st_pt, st_x, ms, mt, noise=noise, graph=False)
prof = lib.prof_collect()
lib.prof_enable(False)
tot = sum(v[1] for v in prof.values())
res["kernel_families"] = {k: {"launches": v[0], "ms": v[1], "share": v[1] / tot} for k, v in prof.items() if v[0]}
c, ms_, work = prof["gemm_split"]
if c:
    ach = work / (ms_ * 1e-3) / 1e12
    res["roofline"] = {"kernel": "pgemm_kernel<9,3> (family gemm_split)", "bound": "mfma", "achieved": ach, "peak": 2516.6 / 3, "unit": "TFLOP/s",
                       "frac": ach / (2516.6 / 3), "avg_us_per_launch": ms_ / c * 1e3, "traffic": None}
c, ms_, work = prof["attention"]
if c:
    res["attention"] = {"kernel": "attention_planes_kernel (d = 132 -> 9 k-chunks per head)", "achieved_TFLOPs": work / (ms_ * 1e-3) / 1e12, "avg_us_per_launch": ms_ / c * 1e3}
# ---- parity beside it: the reference-minted 512 x 512 fixture (one pair, masks 470 / 391, seed 62) through the same engine
g = np.load(os.path.join(ROOT, "tests", "golden", "4dmatch_loop_n512_s20_mc40_masked.npz"))
pr = synth.make_pair(N, M, v["C"], seed=62)
T1 = lambda k: torch.from_numpy(pr[k])[None].cuda()
o = eng.run(T1("src_feats"), T1("tgt_feats"), T1("s_pcd"), T1("t_pcd"), T1("x_T"), (torch.arange(N)[None] < 470).cuda(), (torch.arange(M)[None] < 391).cuda(),
            noise=torch.from_numpy(synth.step_noise(N, M, 62, steps))[:, None].cuda(), trace=True)
res["parity_vs_reference_fixture"] = {"fixture": "tests/golden/4dmatch_loop_n512_s20_mc40_masked.npz",
                                      "max_abs_R_forwd": float(np.abs(o["R_forwd"][:, 0].cpu().numpy() - g["R_forwd"]).max()),
                                      "max_abs_t_forwd": float(np.abs(o["t_forwd"][:, 0].cpu().numpy() - g["t_forwd"]).max()),
                                      "max_abs_conf": float(np.abs(o["conf_matrix_pred"][0].cpu().numpy() - g["conf"]).max()),
                                      "tolerance": "1e-4 outside the committed exemption list (tests/test_loop_gpu.py::test_cfg3_4dmatch_512_batch8_20_steps)"}
print(json.dumps(res))
