#!/usr/bin/env python3
"""bench.py -- scene-pairs/s of the reverse-diffusion matching loop on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: the whole 20-denoise-step
loop (3D/models/pipeline.py:221-283) for `--pairs` independent scene pairs (each a B = 1 problem of
the reference), N = M = 256 superpoints, C = 432, warp active (max_condition_num = 200),
inputs resident in HBM, launched as one HIP-graph replay.  Pairs shard across ranks with no
data-path collective (weak scaling); RCCL is used only for the barrier / max-time / checksum gather.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel family (by GPU time, HIP events on the launch stream) vs its peak,
                and the batched Sinkhorn micro-benchmark vs the 8 TB/s HBM peak (north_star)
  cpu_baseline  the oracle (PyTorch CPU restatement, "port") timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HEAD_GAIN = 3.0                     # the matching head at a checkpoint-like scale (logits O(10)): the "soft" family of tests/ -- the timed engine and the primary
                                    # parity sample (the head's gain scales ONE weight matrix: the work per pair does not depend on it)
HEAD_GAIN_STRESS = 24.0             # the stress head of the main fixture family (logits in the thousands: one float32 ulp of a logit is ~1e-4 of x_start): second parity block
PEAK_MFMA_F32_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense fp32-input MFMA
KERNEL_BOUNDARY_US = 1.5             # a dependent kernel boundary on this part (MI355X_MICROARCH.md, price list row "boundary": 1.1-1.9 us):
                                     # the floor a chain of n launches is priced against
DEP_LAUNCH_FLOOR_US = 4.7            # the cheapest kernels of the single-pair chain (LayerNorm of 512 rows, casts, an attention launch
                                     # that returns at once): 4.7-4.9 us each; an empty kernel in an idle chain: 1.53 us
                                     # (profiles/r02_b1_launch_floor.json, DESIGN section 5 "Single pair")
PEAK_MFMA_BF16_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF)
# the split-operand GEMMs spend three 16-bit MFMA MACs per fp32 MAC: the ceiling of what they execute, in fp32-equivalent FLOP/s.
# The layer GEMMs (dr::pgemm_kernel, csrc/pgemm.hip) run on fp16 hi / lo plane images of both operands: three fp16 MFMA products per
# fp32 MAC (the fp16 and bf16 dense MFMA peaks are the same).
SPLIT_PRODUCTS = 3.0
PEAK_SPLIT_TFLOPS = PEAK_MFMA_BF16_TFLOPS / SPLIT_PRODUCTS
SPLIT_KERNEL = "pgemm_kernel"
SPLIT_TEXT = ("three fp16 MFMA products of hi/lo operand planes (rows of both operands scaled by exact powers of two into fp16's range; "
              "planes written by the producing kernels, both operands streamed by LDS-DMA)")
LOOP_PMC = os.path.join(ROOT, "profiles", "r06_pgemm_loop_pmc.json")   # rocprofv3 --pmc passes over the loop's own launches (tools/pmc_collect.py)
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
METRIC = "scene-pairs/sec @ 20 denoise steps (N=M=256); IR/FMR parity vs ref"


def make_inputs(variant, P, N, M, seed0, device, seeds=None):
    from diffreg_hip import synth
    C = synth.VARIANTS[variant]["C"]
    prs = [synth.make_pair(N, M, C, seed=sd) for sd in (seeds if seeds is not None else range(seed0, seed0 + P))]
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(device)
    return prs, dict(f_s=st("src_feats"), f_t=st("tgt_feats"), p_s=st("s_pcd"), p_t=st("t_pcd"), x_T=st("x_T"))


def make_engine(variant, steps, mc, device, strict_f64=False, head_gain=HEAD_GAIN):
    from diffreg_hip import synth
    from diffreg_hip.engine import DenoiseEngine
    v = synth.VARIANTS[variant]
    W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=head_gain).items()}
    return W, DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps,
                            sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=mc,
                            n_layers=v["n_layers"], device=device, strict_f64=strict_f64)


def sinkhorn_microbench(device, B=4096, N=256, M=256, reps=20):
    from diffreg_hip import lib
    x = torch.randn(B, N, M, device=device) * 2
    a = torch.tensor(1.0, device=device)
    out = lib.sinkhorn(x, a, 3)
    # warm-up by time, not by count: the first launches of a process run on a chip that is still ramping its clocks (the same
    # 20 launches measure 428 us right after start-up and 400 us a moment later)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3:
        for _ in range(10):
            lib.sinkhorn(x, a, 3, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.sinkhorn(x, a, 3, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = B * N * M * 8
    # the same bytes as a plain device copy (torch's copy kernel), for context
    y = torch.empty_like(x)
    for _ in range(2):
        y.copy_(x)
    e0.record()
    for _ in range(10):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = byts / (e0.elapsed_time(e1) / 10) / 1e6
    del y
    # latency of one tile
    x1 = x[:1].contiguous()
    o1 = lib.sinkhorn(x1, a, 3)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        lib.sinkhorn(x1, a, 3, out=o1)
    e1.record()
    torch.cuda.synchronize()
    # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 correction + WRITE_SIZE)
    traffic, src = None, None
    pmc = os.path.join(ROOT, "profiles", "r03_sinkhorn_persist_pmc_traffic.json")
    if os.path.exists(pmc) and B == 4096 and N == 256 and M == 256:
        traffic, src = json.load(open(pmc))["hbm_bytes_per_launch"], "profiles/r03_sinkhorn_persist_pmc_traffic.json (python3 tools/sk_one.py 4096; tools/pmc_collect.py)"
    return dict(kernel="sk_fast_persist_kernel (>= 512 tiles; sk_fast_kernel<float,float,16,4> below)", bound="hbm",
                device_copy_same_bytes_GBps=copy_gbps, tiles_per_launch=B, bytes_per_tile=N * M * 8,
                us_per_launch=ms * 1e3, achieved=byts / ms / 1e6, peak=PEAK_HBM_GBPS, unit="GB/s",
                frac=byts / ms / 1e6 / PEAK_HBM_GBPS, traffic=traffic, traffic_source=src, algorithmic_bytes=byts,
                single_tile_latency_us=e0.elapsed_time(e1) / 50 * 1e3)


def _kth_gap_rel(conf, K):
    """relative gap between the K-th and the (K+1)-th largest entry of a confidence tile (the boundary of the Procrustes top-K)"""
    v = conf.reshape(-1).topk(K + 1)[0]
    return float((v[K - 1] - v[K]) / v[K - 1])


def _oracle_pair(W, v, variant, N, M, steps, mc, seed, f64=False):
    """one oracle loop on the host -> (seconds, record of what the parity sample compares: match list, IR, conf, per-step pose / cond / K-th gap)"""
    from diffreg_hip import synth
    from oracle import diffreg_oracle as orc
    p = synth.make_pair(N, M, v["C"], seed=seed)
    T = lambda a: torch.from_numpy(a)[None]
    up = (lambda t: t.double()) if f64 else (lambda t: t)
    ms, mt = torch.ones(1, N, dtype=torch.bool), torch.ones(1, M, dtype=torch.bool)
    tr = []
    t0 = time.perf_counter()
    o = orc.denoise_loop(W, v, up(T(p["src_feats"])), up(T(p["tgt_feats"])), T(p["s_pcd"]), T(p["t_pcd"]), ms, mt, up(T(p["x_T"])),
                         steps, mc, variant=variant, trace=tr)
    dt = time.perf_counter() - t0
    K = int(max(N, M) * v["sample_rate"])
    rec = dict(seed=seed, match_pred=o["match_pred"], conf=o["conf_matrix_pred"][0].clone(),
               ir=orc.inlier_ratio(o["match_pred"], T(p["s_pcd"]), T(p["t_pcd"]), p["R_gt"], p["t_gt"]),
               R_forwd=torch.stack([r["R_forwd"][0] for r in tr]).float(), cond=[float(r["cond"][0]) for r in tr],
               kth_gap_rel=[_kth_gap_rel(r["conf"][0], K) for r in tr])
    return dt, rec


def cpu_baseline(variant, N, M, steps, mc, budget_s=25.0):
    """oracle loop on the host cores: 1 warm-up pair + as many timed pairs as fit the budget (>= 1)."""
    from diffreg_hip import synth
    # the loop is ~1 500 small torch ops per pair: more threads than ~16 only add synchronisation cost (measured on the GPU box: 8 -> 1.26, 16 -> 1.46, 32 -> 0.77, 64 -> 0.33 pairs/s)
    # (256 threads on the 256-CPU GPU box are > 100x slower than 32)
    cores = int(os.environ.get("DIFFREG_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    v = synth.VARIANTS[variant]
    W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=HEAD_GAIN).items()}
    kept = []                  # records of the timed pairs: the parity sample
    t_start = time.perf_counter()
    times = [_oracle_pair(W, v, variant, N, M, steps, mc, 1000)[0]]                  # warm-up pair
    if time.perf_counter() - t_start < budget_s:
        times = []
    while not times or (time.perf_counter() - t_start < budget_s and len(times) < 10):
        dt, rec = _oracle_pair(W, v, variant, N, M, steps, mc, 1001 + len(kept))
        times.append(dt); kept.append(rec)
    med = float(np.median(times))
    return dict(value=1.0 / med, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample="%d pairs of N=M=%d, %d denoise steps after 1 warm-up pair (median %.3f s/pair); "
                       "oracle/diffreg_oracle.py on torch %s CPU" % (len(times), N, steps, med, torch.__version__)), kept


_cpu_baseline_orig = cpu_baseline


def ir_fmr_parity(eng, variant, N, M, kept, device, head_gain):
    """The metric's parity leg: inlier ratio (3D/models/loss.py:383-410, thr 0.1) and feature-matching recall
    (IR > 0.05, 3D/lib/tester.py:83-85) of the HIP loop's match_pred against the oracle's on the same synthetic pairs
    (ground-truth pose of the generator), and -- pair by pair -- whether the two FREE-RUNNING trajectories stay within the 1e-4 contract
    (pairs_within_1e4).  A pair that does not is examined at the first step whose R_forwd differs: the relative gap between the K-th and
    the (K+1)-th warp confidence (the top-K boundary of procrustes.py:66) and the distance of cond from the gate in BOTH runs, and the same
    comparison between two runs of the oracle itself (float32 against float64 weights / features / state).  What the free-running sample
    cannot show for such a pair -- the steps behind the divergence -- is asserted step by step in tests/test_teacher_forced_gpu.py."""
    from diffreg_hip import synth
    from oracle import diffreg_oracle as orc
    seeds = [k["seed"] for k in kept]
    prs, inp = make_inputs(variant, len(seeds), N, M, seed0=seeds[0], device=device)
    assert seeds == list(range(seeds[0], seeds[0] + len(seeds)))
    out = eng.run(inp["f_s"], inp["f_t"], inp["p_s"], inp["p_t"], inp["x_T"], graph=False, trace="full")
    torch.cuda.synchronize()
    ml = eng.match_list(out)
    v = synth.VARIANTS[variant]
    K = int(max(N, M) * v["sample_rate"])
    gate = float(eng.cfg.max_condition_num)
    # the same matches through the device harness (dr_inlier_ratio_f32, dr_ransac_corr_f64): IR on the device,
    # and the pose a 50 000-hypothesis correspondence RANSAC recovers from them, against the generator's ground truth
    from diffreg_hip import metrics as dmet
    rot_gt = torch.tensor(np.stack([p["R_gt"] for p in prs]), dtype=torch.float32, device=device)
    trn_gt = torch.tensor(np.stack([p["t_gt"] for p in prs]), dtype=torch.float32, device=device)
    ev = dmet.evaluate_pairs(out["matches_padded"], out["match_count"], inp["p_s"], inp["p_t"], rot_gt, trn_gt,
                             pair_ids=torch.tensor(seeds))
    ir_dev = ev["ir"].cpu().numpy()
    rot_err = (ev["rot"].float() - rot_gt).abs().amax(dim=(1, 2)).cpu().numpy()
    trn_err = (ev["trn"][:, :, 0].float() - trn_gt).abs().amax(dim=1).cpu().numpy()
    ir_hip, ir_ref, jac, per_pair, within = [], [], [], [], 0
    W64 = None
    for i, rec in enumerate(kept):
        p = prs[i]
        mh = ml[i].cpu()
        ir_hip.append(orc.inlier_ratio(mh, torch.from_numpy(p["s_pcd"])[None], torch.from_numpy(p["t_pcd"])[None], p["R_gt"], p["t_gt"]))
        ir_ref.append(rec["ir"])
        a, b = set(map(tuple, mh[:, 1:].tolist())), set(map(tuple, rec["match_pred"][:, 1:].tolist()))
        jac.append(len(a & b) / max(1, len(a | b)))
        # where the two runs part, if they do: per-step pose deviation (the trajectory), the final matrix, and how DECIDED the
        # read-out's arg-maxima are (a column of an unmatched target holds nearly equal entries: margins below 1e-7 -- its
        # arg-maximum, hence its match-list entry, is noise in ANY float32 evaluation)
        dR = (out["R_forwd"][:, i].cpu() - rec["R_forwd"]).abs().amax(dim=(1, 2))
        first = int(torch.nonzero(dR > 1e-4)[0]) if bool((dR > 1e-4).any()) else None
        dconf = float((out["conf_matrix_pred"][i].cpu() - rec["conf"]).abs().max())
        c = rec["conf"].numpy()
        sr, sc = np.sort(c, 1), np.sort(c, 0)
        tol = 10.0 * max(dconf, 1e-12)
        und_r, und_c = set(np.nonzero(sr[:, -1] - sr[:, -2] <= tol)[0].tolist()), set(np.nonzero(sc[-1] - sc[-2] <= tol)[0].tolist())
        diff = a ^ b
        decided_diff = [e for e in diff if not (e[0] in und_r or e[1] in und_c)]
        ok = first is None and dconf <= 1e-4
        within += int(ok)
        ent = dict(seed=rec["seed"], within_1e4=ok, jaccard=jac[-1], max_abs_dconf=dconf, max_abs_dR_forwd=float(dR.max()), first_step_R_differs_1e4=first,
                   matches_hip=len(a), matches_oracle=len(b), differing_entries=len(diff),
                   undecided_rows=len(und_r), undecided_columns=len(und_c), differing_entries_at_decided_argmaxima=len(decided_diff))
        if not ok:
            k0 = first if first is not None else eng.steps - 1
            cond_h = out["cond"][:, i].cpu().tolist()
            ent["at_first_divergent_step"] = dict(
                step=k0, kth_gap_rel_oracle=rec["kth_gap_rel"][k0], kth_gap_rel_hip=_kth_gap_rel(out["wconf"][k0, i], K),
                min_kth_gap_rel_oracle_up_to_here=min(rec["kth_gap_rel"][:k0 + 1]),
                cond_oracle=rec["cond"][k0], cond_hip=cond_h[k0], gate=gate,
                abs_cond_minus_gate_oracle=abs(rec["cond"][k0] - gate), abs_cond_minus_gate_hip=abs(cond_h[k0] - gate))
            # the control, on EVERY diverging pair: the same comparison between two runs of the oracle itself, float32 vs float64
            try:
                if W64 is None:
                    W64 = {k_: torch.from_numpy(a_).double() for k_, a_ in synth.make_weights(v["C"], seed=7, head_gain=head_gain).items()}
                _, r64 = _oracle_pair(W64, v, variant, N, M, eng.steps, gate, rec["seed"], f64=True)
                b64 = set(map(tuple, r64["match_pred"][:, 1:].tolist()))
                dR64 = (r64["R_forwd"] - rec["R_forwd"]).abs().amax(dim=(1, 2))
                ent["control_oracle_f32_vs_f64"] = dict(
                    jaccard=len(b & b64) / max(1, len(b | b64)), max_abs_dconf=float((rec["conf"] - r64["conf"]).abs().max()),
                    max_abs_dR_forwd=float(dR64.max()), first_step_R_differs_1e4=int(torch.nonzero(dR64 > 1e-4)[0]) if bool((dR64 > 1e-4).any()) else None)
            except Exception as e:
                ent["control_oracle_f32_vs_f64"] = {"error": "%s: %s" % (type(e).__name__, e)}
        per_pair.append(ent)
    ir_hip, ir_ref = np.array(ir_hip), np.array(ir_ref)
    return dict(pairs=len(kept), head_gain=head_gain, pairs_within_1e4=within,
                pairs_within_1e4_what="pairs whose free-running HIP trajectory stays within 1e-4 of the oracle's on R_forwd at every step AND on every entry of conf_matrix_pred",
                ir_hip=float(ir_hip.mean()), ir_oracle=float(ir_ref.mean()),
                max_abs_ir_diff=float(np.abs(ir_hip - ir_ref).max()), fmr_hip=float((ir_hip > 0.05).mean()),
                fmr_oracle=float((ir_ref > 0.05).mean()), match_set_jaccard_min=float(min(jac)),
                per_pair=per_pair,
                match_set_note="a match list is row arg-maxima united with column arg-maxima; columns of unmatched targets hold nearly equal "
                               "entries (margins < 1e-7), so their arg-maxima differ between any two float32 evaluations -- see the per-pair controls; "
                               "differing_entries_at_decided_argmaxima counts the differences that are NOT of that kind (expected 0)",
                ir_hip_device_kernel=float(ir_dev.mean()), max_abs_ir_device_vs_host=float(np.abs(ir_dev - ir_hip).max()),
                ransac_50000=dict(max_abs_R_err=float(rot_err.max()), max_abs_t_err=float(trn_err.max()),
                                  mean_fitness=float(ev["fitness"].mean())),
                tolerance="IR / FMR within 0.1 (north_star); trajectories 1e-4", note="synthetic scenes, generator ground truth; oracle = CPU restatement pinned to the reference")


def stress_head_parity(variant, N, M, steps, mc, seeds, device):
    """the same sample on the STRESS head (HEAD_GAIN 24: matching logits in the thousands, where top-K near-ties flip between any two float32
    evaluations): a second engine + a second oracle pass over the same seeds"""
    from diffreg_hip import synth
    v = synth.VARIANTS[variant]
    W, eng = make_engine(variant, steps, mc, device, head_gain=HEAD_GAIN_STRESS)
    kept = [_oracle_pair(W, v, variant, N, M, steps, mc, sd)[1] for sd in seeds]
    return ir_fmr_parity(eng, variant, N, M, kept, device, HEAD_GAIN_STRESS)


def pgemm_per_op(dev, rows, C=432, reps=10):
    """The four launches of one GeometryAttentionLayer call of the headline's batch (rows = token rows of one batch, C = 432) on their own: whole-launch
    time (HIP events) and fraction of the three-product ceiling per op -- where the family's average comes from (q | k | v and mlp0
    end in a plane-image epilogue, merge and mlp2 in the LayerNorm one)."""
    from diffreg_hip import lib
    g = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x = rnd(rows, C)
    img, bnd = lib.planes_from_f32(x)
    msg_img, msg_b = lib.planes_from_f32(rnd(rows, C))
    hid_img, hid_b = lib.planes_from_f32(rnd(rows, 2 * C))
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    lnb = lib.ln_bound(g1, b1)
    o_img = torch.zeros_like(img); o_b = torch.zeros(rows, device=dev); o32 = torch.empty(rows, C, device=dev)
    h_img = torch.zeros_like(hid_img); h_b = torch.zeros(rows, device=dev)
    nbytes = lib.raw().dr_plane_image_bytes(rows, C)
    q_img = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    ang = torch.rand(rows, C // 2, device=dev); cosT, sinT = ang.cos().contiguous(), ang.sin().contiguous()
    pk3 = lib.pack_weight_planes(rnd(3 * C, C) / C ** 0.5, 3, C)
    pk1 = lib.pack_weight_planes(rnd(C, C) / C ** 0.5, 1, C)
    pk0 = lib.pack_weight_planes(rnd(2 * C, 2 * C) / (2 * C) ** 0.5, 2, C)
    pk2 = lib.pack_weight_planes(rnd(C, 2 * C) / (2 * C) ** 0.5, 1, C)
    o3 = torch.empty(rows, 3 * C, device=dev)
    ops = {
        "q|k|v (fp32 rows + rotary; the loop writes plane images)": (lambda: lib.linear_planes(rows, C, 3, img, bnd, C, pk3, lib.PL_F32, out=o3, ldo=3 * C, blk_stride=C, cos_t=cosT, sin_t=sinT, rot_mask=3, rot_C=C), 3 * C * C),
        "merge + LayerNorm": (lambda: lib.linear_planes(rows, C, 1, img, bnd, C, pk1, lib.PL_LN, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, lnb=lnb), C * C),
        "mlp0 + ReLU": (lambda: lib.linear_planes(rows, C, 2, img, bnd, C, pk0, lib.PL_PLANES, a1=msg_img, b1=msg_b, k1=C, out_image=h_img, out_image_k=2 * C, out_bound=h_b, relu=True), 4 * C * C),
        "mlp2 + LayerNorm + residual": (lambda: lib.linear_planes(rows, C, 1, hid_img, hid_b, 2 * C, pk2, lib.PL_LN, out=o32, ldo=C, out_image=o_img, out_image_k=C, out_bound=o_b, gamma=g1, beta=b1, resid=x, ldr=C, bound_resid=bnd, lnb=lnb), 2 * C * C)}
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, (fn, kn) in ops.items():
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tf = 2.0 * rows * kn / us / 1e6
        out[name] = {"us_per_launch": us, "TFLOPs": tf, "frac": tf / PEAK_SPLIT_TFLOPS}
    return {"rows": rows, "C": C, "ops": out, "measured": "stand-alone launches of the four shapes through dr_linear_planes_f32 (HIP events, %d launches each)" % reps}


def _families(prof):
    tot = sum(v[1] for v in prof.values()) or 1.0
    return {k: {"launches": v[0], "ms": v[1], "share": v[1] / tot} for k, v in prof.items() if v[0]}


def _pmc(name):
    """a committed rocprofv3 --pmc summary of a kernel inside one of the other configurations (tools/pmc_collect.py; tools/experiments/_prof_r05_pmc_cfg.sh)"""
    f = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    return {"hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"), "avg_us_per_launch_pmc_run": d.get("avg_us_per_launch_profiled"),
            "mfma_busy_fraction_of_wall": d.get("derived", {}).get("mfma_busy_fraction_of_wall"),
            "wave_parked_fraction": d.get("derived", {}).get("wave_parked_fraction"), "source": "profiles/" + name}


def _mfma_roofline(prof, kernel_split, kernel_attn, pmc_split=None, pmc_attn=None):
    """roofline objects of the two MFMA families of a profiled call: the plane GEMM against the 3-product fp16 ceiling, attention against
    the ceiling of the pipe it runs on (plane attention: the same 3-product ceiling; the f32-input kernels: the f32 MFMA peak)"""
    out = {}
    c, ms_, work = prof.get("gemm_split", (0, 0.0, 0.0))
    if c:
        ach = work / (ms_ * 1e-3) / 1e12
        pm = _pmc(pmc_split) if pmc_split else None
        out["roofline"] = {"kernel": kernel_split, "bound": "mfma", "achieved": ach, "peak": PEAK_SPLIT_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_SPLIT_TFLOPS, "avg_us_per_launch": ms_ / c * 1e3, "traffic": pm["hbm_bytes_per_launch"] if pm else None}
        if pm:
            out["roofline"]["pmc"] = pm
    c, ms_, work = prof.get("attention", (0, 0.0, 0.0))
    if c:
        ach = work / (ms_ * 1e-3) / 1e12
        out["attention"] = {"kernel": kernel_attn, "bound": "mfma", "achieved": ach, "unit": "TFLOP/s", "avg_us_per_launch": ms_ / c * 1e3}
        if pmc_attn and _pmc(pmc_attn):
            out["attention"]["pmc"] = _pmc(pmc_attn)
    c, ms_, work = prof.get("gemm", (0, 0.0, 0.0))
    if c:
        ach = work / (ms_ * 1e-3) / 1e12
        out["gemm_f32"] = {"kernel": "gemm_nt_* (f32-input MFMA)", "bound": "mfma", "achieved": ach, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_MFMA_F32_TFLOPS, "avg_us_per_launch": ms_ / c * 1e3}
    return out


def _time_calls(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def bench_cfg3(dev):
    """BASELINE configs[2]: 4DMatch N = M = 512, C = 528 (d_head 132), 20 denoise steps, batch of 8 pairs per call (masks all-true, sigma*xi
    active): one captured call on one stream, and two such calls on two streams (how cfg2's headline fills the chip)."""
    from diffreg_hip import lib, synth
    from diffreg_hip.engine import DenoiseEngine
    variant, N, M, steps, mc, P = "4dmatch", 512, 512, 20, 40.0, 8
    v = synth.VARIANTS[variant]
    W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=HEAD_GAIN).items()}
    eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                        sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=dev)

    def group(seed0):
        prs = [synth.make_pair(N, M, v["C"], seed=seed0 + i) for i in range(P)]
        st = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev)
        noise = torch.from_numpy(np.stack([synth.step_noise(N, M, seed0 + i, steps) for i in range(P)], 1)).to(dev)
        return dict(src_feats=st("src_feats"), tgt_feats=st("tgt_feats"), s_pcd=st("s_pcd"), t_pcd=st("t_pcd"), x_T=st("x_T"),
                    src_mask=torch.ones(P, N, dtype=torch.bool, device=dev), tgt_mask=torch.ones(P, M, dtype=torch.bool, device=dev), noise=noise)
    g0, g1 = group(300), group(320)
    one = _time_calls(lambda: eng.run(graph=True, borrow=True, **g0), warm=3, reps=5)
    two = _time_calls(lambda: eng.run_streams([g0, g1], 2), warm=3, reps=5)
    three = None
    if os.environ.get("DIFFREG_BENCH_CFG3_STREAMS", "3") == "3":
        g2 = group(340)
        three = _time_calls(lambda: eng.run_streams([g0, g1, g2], 3), warm=3, reps=4)
    res = {"workload": "cfg3: 4DMatch N=M=512, C=528 (d_head 132), %d denoise steps, %d pairs per call, max_condition_num=%g" % (steps, P, mc),
           "ms_per_call": one * 1e3, "pairs_per_s": P / one,
           "two_concurrent_calls": {"ms_per_pass": two * 1e3, "pairs_per_s": 2 * P / two,
                                    "what": "two independent 8-pair calls, one captured graph each, on two HIP streams (16 pairs in flight)"}}
    if three:
        res["three_concurrent_calls"] = {"ms_per_pass": three * 1e3, "pairs_per_s": 3 * P / three, "what": "three 8-pair calls on three HIP streams (24 pairs in flight)"}
    eng.run(graph=False, borrow=True, **g0)
    torch.cuda.synchronize()
    lib.prof_enable(True)
    eng.run(graph=False, borrow=True, **g0)
    prof = lib.prof_collect()
    lib.prof_enable(False)
    res["kernel_families"] = _families(prof)
    res.update(_mfma_roofline(prof, "pgemm16w_kernel (128 x 288 wide-wave workgroups) + pgemm_kernel<9,3,LN,2> (64-row, k-split)", "attention_planes_kernel<9,5> (d = 132)",
                              "r06_cfg3_pgemm_pmc.json", "r06_cfg3_attention_pmc.json"))
    res["measured_on"] = "one eager 8-pair call (HIP events on the launch stream)"
    # the OPT-IN reduced-precision attention (DR_LOOP_ATTN_F16: one fp16 product per contraction, BASELINE's "bf16 MFMA attention"): rate and deviation
    ref_conf = eng.run(graph=False, **g0)["conf_matrix_pred"].clone()
    eng16 = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                          sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=dev, attn_f16=True)
    t16 = _time_calls(lambda: eng16.run(graph=True, borrow=True, **g0), warm=3, reps=5)
    c16 = eng16.run(graph=False, **g0)["conf_matrix_pred"]
    res["opt_in_attn_f16"] = {"ms_per_call": t16 * 1e3, "pairs_per_s": P / t16, "max_abs_dconf_vs_default": float((c16 - ref_conf).abs().max()),
                              "mean_abs_dconf_vs_default": float((c16 - ref_conf).abs().mean()), "conf_range": "[0, 1] (sigmoid read-out)",
                              "note": "outside the 1e-4 contract by design; never the default"}
    return res


def _jaccard(a, b, i):
    ca, cb = int(a["match_count"][i]), int(b["match_count"][i])
    sa = set(map(tuple, a["matches_padded"][i, :ca, 1:].cpu().tolist())); sb = set(map(tuple, b["matches_padded"][i, :cb, 1:].cpu().tolist()))
    return len(sa & sb) / max(1, len(sa | sb))


def bench_cfg5(dev, batches=(1, 8)):
    """BASELINE configs[4]: 2D-3D, N = 1024 point nodes x M = 2048 image patches, C = 256, 10 denoise steps, warp active: one pair per call
    (f32-input MFMA kernels) and 8 pairs per call (plane-image path: 24 576 token rows)."""
    from diffreg_hip import lib, synth
    from diffreg_hip.engine import DenoiseEngine2D3D
    N, M, steps, mc = 1024, 2048, 10, 200.0
    Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
    W = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in Wn.items()}
    eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=dev)
    distinct = [synth.make_pair_2d3d(N, M, 60 + i, weights=Wn) for i in range(min(4, max(batches)))]
    res = {"workload": "cfg5: 2D-3D N=1024 x M=2048, C=256 (d_head 64), %d denoise steps, max_condition_num=%g" % (steps, mc), "per_batch": {}}
    for P in batches:
        prs = [distinct[i % len(distinct)] for i in range(P)]
        args = [torch.from_numpy(np.stack([p[k] for p in prs])).to(dev) for k in ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")]
        kw = dict(zip(eng._ARGS, args))
        dt = _time_calls(lambda: eng.run_static(slot=0, graph=True, **kw), warm=3, reps=4)
        ent = {"ms_per_call": dt * 1e3, "pairs_per_s": P / dt, "ms_per_pair": dt * 1e3 / P, "launch": "one captured HIP graph per call",
               "distinct_scenes": min(P, len(distinct)), "path": "plane images" if P * (N + M) >= 4096 else "f32-input MFMA kernels"}
        if P > 1:
            two = _time_calls(lambda: eng.run_streams([kw, kw], 2), warm=3, reps=4)
            ent["two_concurrent_calls"] = {"ms_per_pass": two * 1e3, "pairs_per_s": 2 * P / two,
                                           "what": "two independent %d-pair calls, one captured graph each, on two HIP streams" % P}
            three = _time_calls(lambda: eng.run_streams([kw, kw, kw], 3), warm=3, reps=4)
            ent["three_concurrent_calls"] = {"ms_per_pass": three * 1e3, "pairs_per_s": 3 * P / three, "what": "three %d-pair calls on three HIP streams" % P}
            ref = {k_: v_.clone() for k_, v_ in eng.run_static(slot=0, graph=True, **kw).items()}
            eng16 = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=dev, attn_f16=True)
            t16 = _time_calls(lambda: eng16.run_static(slot=0, graph=True, **kw), warm=3, reps=4)
            o16 = eng16.run_static(slot=0, graph=True, **kw)
            torch.cuda.synchronize()
            xs = lambda o_: torch.nan_to_num(o_["x_final"], neginf=0.0)
            ent["opt_in_attn_f16"] = {"ms_per_call": t16 * 1e3, "pairs_per_s": P / t16,
                                      "max_abs_dconf_vs_default": float((o16["conf_matrix_pred"] - ref["conf_matrix_pred"]).abs().max()),
                                      "conf_max": float(ref["conf_matrix_pred"].max()),
                                      "max_abs_dstate_vs_default": float((xs(o16) - xs(ref)).abs().max()),
                                      "match_list_jaccard_min": min(_jaccard(o16, ref, i_) for i_ in range(P)),
                                      "note": "DR_LOOP_ATTN_F16: one fp16 product per contraction in q k^T and P v (BASELINE's 'fp16 MFMA cross-attn'); "
                                              "outside the 1e-4 contract by design; never the default"}
            del eng16
        lib.prof_enable(True)
        eng.run(*args)
        prof = lib.prof_collect()
        lib.prof_enable(False)
        ent["kernel_families"] = _families(prof)
        ent.update(_mfma_roofline(prof, "pgemm_kernel<4,4> (256-column geometry; bias / post-add LayerNorm epilogues)",
                                  "attention_planes_kernel<4,2> (d = 64)" if P * (N + M) >= 4096 else "attention_kernel / attention_flash_kernel (f32-input MFMA, d = 64)",
                                  "r05_cfg5_pgemm_pmc.json" if P * (N + M) >= 4096 else None, "r05_cfg5_attention_pmc.json" if P * (N + M) >= 4096 else None))
        if "roofline" in ent and P * (N + M) >= 4096:
            # The same family against the OTHER roof.  At C = 256 a layer call's four plane GEMMs move 16 matrix passes of rows x 256 x 4 B (q | k | v images,
            # fp32 residual rows + their images, the 512-wide hidden image) for 8 GEMM units of 2 rows 256^2 FLOP: ~64 FLOP/B over the family (32 for the lin
            # launch, 96 for q | k | v), under the ridge point of the three-product ceiling (105 FLOP/B) -- the family's nearer roof is HBM.  Algorithmic bytes of the family per call from the layer schedule
            # (self, cross) x 3 per step; the image half of layer 0 once per call:
            C4 = 256 * 4
            def layer_bytes(R, Ry=None):      # x rows R; key / value rows Ry (None: self attention, the same rows)
                return (16 * R if Ry is None else 14 * R + 3 * Ry) * C4
            per_step = 2 * layer_bytes(P * (N + M)) + layer_bytes(P * N) + 3 * (layer_bytes(P * M, P * N) + layer_bytes(P * N, P * M))
            total = steps * per_step + layer_bytes(P * M)
            ms_fam = prof["gemm_split"][1]
            gbps = total / (ms_fam * 1e-3) / 1e9
            rf = ent["roofline"]
            rf["other_roof"] = {"bound": "hbm", "algorithmic_bytes_per_call": total, "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                                "flop_per_byte": prof["gemm_split"][2] / total,
                                "note": "activation traffic of the family's launches from the layer schedule (images in / out, fp32 residual rows); weights not counted"}
            if rf["other_roof"]["frac"] > rf["frac"]:
                rf["nearer_roof"] = "hbm"
        res["per_batch"]["P%d" % P] = ent
    best = max(res["per_batch"].values(), key=lambda e: e["pairs_per_s"])
    res["pairs_per_s"] = max(best["pairs_per_s"], best.get("two_concurrent_calls", {}).get("pairs_per_s", 0.0), best.get("three_concurrent_calls", {}).get("pairs_per_s", 0.0))
    if "roofline" in best:
        res["roofline"] = best["roofline"]
    elif "gemm_f32" in best:
        res["roofline"] = best["gemm_f32"]
    return res


def bench_b1_real_size(dev):
    """the literal inference mode of the reference's tester (B = 1, 3D/lib/tester.py:115) at a real 3DMatch pair's coarse size
    (564 x 629 superpoints): latency of one 20-step loop, graph replay"""
    from diffreg_hip import lib, synth
    from diffreg_hip.engine import DenoiseEngine
    variant, N, M, steps, mc = "3dmatch", 564, 629, 20, 200.0
    v = synth.VARIANTS[variant]
    W = {k: torch.from_numpy(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=HEAD_GAIN).items()}
    eng = DenoiseEngine(W, variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                        sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=dev)
    p = synth.make_pair(N, M, v["C"], seed=9000)
    a = [torch.from_numpy(p[k])[None].to(dev) for k in ("src_feats", "tgt_feats", "s_pcd", "t_pcd", "x_T")]
    dt = _time_calls(lambda: eng.run(*a, graph=True, borrow=True), warm=3, reps=10)
    lib.prof_enable(True)
    eng.run(*a, graph=False, borrow=True)
    prof = lib.prof_collect()
    lib.prof_enable(False)
    n = sum(v_[0] for v_ in prof.values())
    fl = sum(v_[2] for k, v_ in prof.items() if k in ("gemm", "gemm_split", "attention"))
    return {"workload": "one 3DMatch pair at its real coarse size (N=564 x M=629, C=432), 20 denoise steps, B = 1, graph replay",
            "ms_per_pair": dt * 1e3, "pairs_per_s": 1.0 / dt, "kernel_families": _families(prof),
            "roofline": {"bound": "dependent-launch chain", "launches": n, "boundary_us": KERNEL_BOUNDARY_US,
                         "floor_ms": n * KERNEL_BOUNDARY_US * 1e-3, "frac": n * KERNEL_BOUNDARY_US * 1e-6 / dt,
                         "mfma_TFLOPs": fl / dt / 1e12, "mfma_frac_of_f32_peak": fl / dt / 1e12 / PEAK_MFMA_F32_TFLOPS}}


MAX_LINE_BYTES = 6144               # the printed line must stay under this (asserted in main(); tests/test_bench_line.py)
DETAILS_NAME = "bench_details.json"


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is a summary; bench_details.json keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def write_details(result):
    """the full record (kernel-family tables, per-pair parity lists with their controls, PMC blocks, notes) -> bench_details.json beside
    bench.py (and under gpurun_out/ when that exists, so that a gpurun call brings it home); returns the path named in the line"""
    path = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, DETAILS_NAME), "w") as f:
                json.dump(result, f, indent=1)
            path = path or os.path.join(os.path.relpath(d, ROOT), DETAILS_NAME).replace("./", "")
        except OSError:
            pass
    return path


def compact_line(result, details_path):
    """the ONE line the driver reads: the contract fields, the dominant kernel's roofline, the Sinkhorn roofline, the CPU baseline, and one number
    + one fraction per other configuration.  Everything else lives in bench_details.json."""
    c = result["config"]
    line = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                   "dtype", "data")}
    line["config"] = _pick(c, ("workload", "pairs_per_pass_per_gpu", "streams", "denoise_steps", "N", "M", "graph", "head_gain", "parallelism"))
    if "roofline" in result:
        rf = result["roofline"]
        line["roofline"] = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_us_per_launch",
                                      "work_per_launch", "measured_on"))
        line["roofline"].setdefault("traffic", None)
        if "mfma_busy_fraction_of_wall_pmc" in rf:
            line["roofline"]["mfma_busy"] = rf["mfma_busy_fraction_of_wall_pmc"]
        if "per_op" in rf:
            line["roofline"]["per_op_frac"] = {k.split(" ")[0]: v["frac"] for k, v in rf["per_op"].items()}
    if "sinkhorn_roofline" in result:
        line["sinkhorn_roofline"] = _pick(result["sinkhorn_roofline"], ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch",
                                                                         "tiles_per_launch", "device_copy_same_bytes_GBps"))
    if "cpu_baseline" in result:
        line["cpu_baseline"] = _pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
    if "single_pair" in result:
        line["single_pair"] = _pick(result["single_pair"], ("ms_per_pair",))
        line["single_pair"]["launches"] = result["single_pair"].get("roofline", {}).get("launches")
    oc = result.get("other_configs")
    if oc:
        o = {}
        for name, e in oc.items():
            if "error" in e:
                o[name] = {"error": e["error"][:160]}
                continue
            ent = _pick(e, ("pairs_per_s", "ms_per_call", "ms_per_pair", "ms_per_step"))
            if name == "cfg5":                     # the 8-pair call on ONE stream is cfg5's number; concurrent calls are labelled fields in the details
                best = e["per_batch"].get("P8") or max(e["per_batch"].values(), key=lambda b: b["pairs_per_s"])
                ent = _pick(best, ("pairs_per_s", "ms_per_call"))
                ent["pairs_per_call"] = 8 if "P8" in e["per_batch"] else None
                if "P1" in e["per_batch"]:
                    ent["one_pair_ms"] = e["per_batch"]["P1"]["ms_per_call"]
                e = best
            rf = e.get("roofline")
            if rf:
                ent["frac"] = rf.get("frac")
                ent["kernel"] = rf.get("kernel", "")[:48]
                ent["bound"] = rf.get("bound")
                if rf.get("nearer_roof") == "hbm":
                    ent["frac_hbm"] = rf["other_roof"]["frac"]
            o[name] = ent
        line["other_configs"] = o
    par = result.get("ir_fmr_parity")
    if par:
        p = _pick(par, ("pairs", "head_gain", "pairs_within_1e4", "ir_hip", "ir_oracle", "max_abs_ir_diff", "fmr_hip", "fmr_oracle", "match_set_jaccard_min"))
        sh = par.get("stress_head")
        if sh:
            p["stress_head"] = {"error": sh["error"][:160]} if "error" in sh else _pick(sh, ("pairs", "head_gain", "pairs_within_1e4", "max_abs_ir_diff", "fmr_hip", "fmr_oracle"))
        line["ir_fmr_parity"] = p
    line["parity_ok"] = result.get("parity_ok")
    mg = result["metric_gather"]
    line["metric_gather"] = _pick(mg, ("backend", "n_pairs", "per_rank_pairs", "per_rank_pairs_per_s", "mean_inlier_ratio", "fmr", "registration_recall") + tuple(
        k for k in mg if k.startswith("sum_")))
    line["conf_checksum"] = result["conf_checksum"]
    line["details"] = details_path
    # (the contract's own top-level numbers keep full precision -- value x ms_per_step must stay consistent; the nested summaries are rounded)
    return {k: (_r(v) if isinstance(v, (dict, list)) else v) for k, v in line.items()}


class _Attr(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _attr(d):
    return _Attr({k: _attr(v) for k, v in d.items()}) if isinstance(d, dict) else d


def bench_train_step(dev):
    """one training step of the WHOLE 3DMatch model on the device -- Pipeline.forward_train (KPFCN backbone's coarse phase on a
    synthetic stacked cloud, both transformers, both matching heads, Procrustes fits) + MatchMotionLoss.forward_train + .backward() into all
    44.9 M parameters (3D/models/pipeline.py:182-216, loss.py:80-170).  B = 1; random-init weights; a made-up ground-truth match list."""
    from diffreg_hip import synth
    from models.loss import MatchMotionLoss
    from models.pipeline import Pipeline
    v = synth.VARIANTS["3dmatch"]
    matching = dict(feature_dim=v["C"], confidence_threshold=0.2, entangled=False, dsmax_temperature=0.1, match_type="sinkhorn", skh_init_bin_score=1.0,
                    skh_iters=3, skh_prefilter=False)
    cfg = _attr(dict(dataset="3dmatch", coarse_matching=matching, SAMPLE_STEP=20,
                     kpfcn_config=dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum", deformable=False,
                                       use_batch_norm=True, fine_feature_dim=264, coarse_level=-2),
                     coarse_transformer=dict(feature_dim=v["C"], n_head=v["H"], layer_types=["self", "cross", "positioning", "self", "cross"],
                                             positioning_type="procrustes", pe_type="rotary", vol_bnds=[list(v["origin"]), [1.093, 0.78, 2.92]],
                                             voxel_size=v["voxel"], feature_matching=dict(matching), entangled=False,
                                             procrustes=dict(max_condition_num=200.0, sample_rate=1.0))))
    torch.manual_seed(0)
    model = Pipeline(cfg)
    with torch.no_grad():                          # the overlay backbone's KPConv parameters start as zeros (they come from a checkpoint): random init
        for name, p_ in model.named_parameters():
            if name.endswith("KPConv.weights"):
                p_.normal_(0.0, (p_.shape[0] * p_.shape[1]) ** -0.5)
            elif name.endswith("KPConv.kernel_points"):
                lvl_extent = 0.03                  # (positions inside the first conv radius; the timing does not depend on them)
                p_.copy_(torch.randn_like(p_) * lvl_extent)
                p_[0].zero_()
    model = model.to(dev).train()
    b = synth.make_kpfcn_batch(n_src=9000, n_tgt=9000, extent=2.0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ns, nt = b["stack_lengths"][2]
    K = min(ns, nt) // 2
    data0 = {k: [T(x) for x in b[k]] for k in ("points", "neighbors", "pools", "upsamples")}
    data0["features"] = T(b["features"])
    data0.update({"src_mask": torch.ones(1, ns, dtype=torch.bool, device=dev), "tgt_mask": torch.ones(1, nt, dtype=torch.bool, device=dev),
                  "src_ind_coarse_split": torch.arange(ns, device=dev), "tgt_ind_coarse_split": torch.arange(nt, device=dev),
                  "src_ind_coarse": torch.arange(ns, device=dev), "tgt_ind_coarse": torch.arange(ns, ns + nt, device=dev),
                  "coarse_matches": [torch.stack([torch.arange(K), (torch.arange(K) * 7) % nt]).to(dev)], "batched_rot": torch.eye(3, device=dev)[None],
                  "batched_trn": torch.zeros(1, 3, 1, device=dev)})
    crit = MatchMotionLoss(dict(focal_alpha=0.25, focal_gamma=2.0, pos_weight=1.0, neg_weight=1.0, motion_loss_type="L1", motion_weight=0.1, match_weight=1,
                                match_type="sinkhorn", positioning_type="procrustes", confidence_threshold_metric=0.05, mutual_nearest=False, inlier_thr=0.1,
                                fmr_thr=0.05, registration_threshold=0.2, dataset="3dmatch"))
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-9)

    def step():
        opt.zero_grad(set_to_none=True)
        info = crit.forward_train(model.forward_train(dict(data0)))
        info["loss"].backward()
        opt.step()
        return float(info["loss"].detach())
    dt = _time_calls(step, warm=2, reps=5)
    with_grad = sum(p.numel() for p in params if p.grad is not None)
    with torch.no_grad():
        model.eval()
        fwd = _time_calls(lambda: model.backbone(dict(data0), phase="coarse"), warm=1, reps=5)
    return {"workload": "one training step, B = 1: %d + %d raw points -> %d x %d superpoints; KPFCN coarse phase + heads, forward_train + loss + backward + SGD step"
                        % (len(b["points"][0]) - b["stack_lengths"][0][1], b["stack_lengths"][0][1], ns, nt),
            "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "parameters_with_gradients": with_grad, "parameters_total": sum(p.numel() for p in params),
            "backbone_forward_only_ms": fwd * 1e3,
            "note": "the 20 GeometryAttentionLayer calls of a step run forward-with-saves and backward as ONE library call each (round 5: 28.5 -> 23.4 ms); "
                    "the backbone's blocks and the matching heads are still driven kernel by kernel from Python; the projections' backward products are "
                    "f32-input MFMA GEMMs on explicitly transposed operands; not a tuned path"}


def other_configs(dev):
    """BASELINE's other configurations inside the driver's own run (short runs, each with its own roofline object): what
    tools/bench_cfg3.py / bench_2d3d.py / bench_e2e.py report at length"""
    out = {}
    for name, fn in (("cfg3", bench_cfg3), ("cfg5", bench_cfg5), ("b1_real_size", bench_b1_real_size), ("train_step", bench_train_step)):
        t0 = time.perf_counter()
        try:
            out[name] = fn(dev)
        except Exception as e:                      # a secondary line must not take the headline down with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out[name]["wall_s_incl_setup"] = time.perf_counter() - t0
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed passes of the hot path (K)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("DIFFREG_BENCH_PAIRS", "256")),
                    help="independent scene pairs per pass and per GPU")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DIFFREG_BENCH_STREAMS", "2")),
                    help="the pairs of a pass are split into this many batches, one captured graph each, replayed "
                         "concurrently on separate HIP streams")
    ap.add_argument("--strict-f64", action="store_true",
                    help="run the Sinkhorn calls on the fp64 state in fp64 arithmetic like the reference (DR_LOOP_STRICT_F64); the default "
                         "computes them in fp32 from the fp64 state (stated in config.state)")
    ap.add_argument("--denoise-steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--max-condition-num", type=float, default=200.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short cfg3 / cfg5 / real-size B = 1 runs (other_configs)")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="shard this many pairs round-robin over the ranks (ranks may differ by one pair: BASELINE configs[3] = 64 "
                         "pairs over 8 GPUs) instead of --pairs per rank")
    ap.add_argument("--cpu-stub", action="store_true",
                    help="TEST ONLY (tests/test_shard_cpu.py): a deterministic CPU stand-in replaces the engine and the device "
                         "harness so that this file's multi-rank control path -- torchrun environment, barrier, max time over "
                         "ranks, the metric-vector all_reduce -- runs under gloo on a box without a GPU; the line it prints "
                         "carries \"data\": \"cpu-stub\" and is not a measurement")
    ap.add_argument("--dist-single-rank", action="store_true",
                    help="with ONE rank, still create the process group (backend nccl = RCCL; env rendezvous on 127.0.0.1) so that the barrier, "
                         "the max over ranks and the metric-vector all_reduce of the multi-GPU path run through RCCL on this GPU "
                         "(tests/test_rccl_gpu.py); not a scaling measurement")
    ap.add_argument("--breakdown-only", action="store_true",
                    help="run only the eager, event-timed passes of one batch (the command profiled with rocprofv3 for profiles/)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = args.cpu_stub
    if args.gpus != world:
        # one process per GPU (the driver launches N ranks with --gpus N; N = 1 runs without a launcher): anything else would
        # report an aggregate over a rank count the line does not name
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: launch one rank per GPU (python -m torch.distributed.run "
                         "--nproc-per-node %d ... bench.py --gpus %d)\n" % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    use_dist = world > 1 or args.dist_single_rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
        if not stub:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo" if stub else "nccl", rank=rank, world_size=world)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    sync = (lambda: None) if stub else torch.cuda.synchronize
    if not stub:
        torch.cuda.set_device(dev)
        from diffreg_hip import lib
    from diffreg_hip import shard
    variant, N, M, S = "3dmatch", args.n, args.n, args.denoise_steps
    # the pairs of this rank: --pairs each (weak scaling), or a round-robin share of --total-pairs (ranks may differ by one)
    pair_ids = shard.shard_pairs(args.total_pairs, rank, world) if args.total_pairs else list(range(rank * args.pairs, (rank + 1) * args.pairs))
    P = len(pair_ids)
    use_graph = not args.no_graph
    nstreams = max(1, min(args.streams, P)) if use_graph else 1
    per = [P // nstreams + (1 if i < P % nstreams else 0) for i in range(nstreams)]
    groups, inp, gt, off = [], None, [], 0
    if stub:
        eng = W = None
    else:
        W, eng = make_engine(variant, S, args.max_condition_num, dev, strict_f64=args.strict_f64)
        for gi, pg in enumerate(per):
            prs_g, ig = make_inputs(variant, pg, N, M, 0, dev, seeds=[5000 + i for i in pair_ids[off:off + pg]])
            off += pg
            inp = inp or ig
            groups.append(dict(src_feats=ig["f_s"], tgt_feats=ig["f_t"], s_pcd=ig["p_s"], t_pcd=ig["p_t"], x_T=ig["x_T"]))
            gt.append((torch.tensor(np.stack([p["R_gt"] for p in prs_g]), dtype=torch.float32, device=dev),
                       torch.tensor(np.stack([p["t_gt"] for p in prs_g]), dtype=torch.float32, device=dev)))

    def run(graph):
        """one pass over this rank's pairs -> list of per-batch outputs"""
        if stub:
            time.sleep(1e-3 * P)                      # stand-in for the loop: time proportional to the rank's pair count
            return [None]
        if graph:
            return eng.run_streams(groups, nstreams)
        return [eng.run(inp["f_s"], inp["f_t"], inp["p_s"], inp["p_t"], inp["x_T"], graph=False)]

    def evaluate(outs):
        """per-pair IR / FMR / RR of the last pass through the device harness -> three [P] tensors"""
        if stub:                                      # deterministic functions of the GLOBAL pair index
            ids = torch.tensor(pair_ids, dtype=torch.float64)
            return (ids % 10) / 10.0, ((ids % 10) / 10.0 > 0.05).double(), (ids % 3 == 0).double()
        from diffreg_hip import metrics as dmet
        from diffreg_hip import synth
        irs, fmrs, rrs, o = [], [], [], 0
        for g_, out_, (Rg, tg) in zip(groups, outs, gt):
            n = Rg.shape[0]
            info = torch.stack([torch.as_tensor(synth.make_info(5000 + i), dtype=torch.float64) for i in pair_ids[o:o + n]]).to(dev)
            ev = dmet.evaluate_pairs(out_["matches_padded"], out_["match_count"], g_["s_pcd"], g_["t_pcd"], Rg, tg, info=info,
                                     pair_ids=torch.tensor(pair_ids[o:o + n]))
            irs.append(ev["ir"].double()); fmrs.append(ev["fmr"].double()); rrs.append(ev["rr_ok"].double())
            o += n
        return torch.cat(irs), torch.cat(fmrs), torch.cat(rrs)

    if args.breakdown_only:
        use_graph = False
        args.no_single_pair = args.no_cpu_baseline = args.no_other_configs = True
    if stub:
        args.no_single_pair = args.no_cpu_baseline = args.no_breakdown = args.no_other_configs = True
    # the Sinkhorn roofline micro-benchmark (batched tiles, HBM-bound) runs first: behind the MFMA-heavy loop the
    # same launch is 8 % slower (the chip is then at its power limit), which would measure the loop's heat, not the kernel
    sk_roof = None
    if rank == 0 and world == 1 and not args.breakdown_only and not stub:
        sk_roof = sinkhorn_microbench(dev)
        sk_roof["measured"] = "before the timed loop (HIP events, 20 launches after 0.3 s of warm-up launches)"
    for _ in range(max(args.warmup, 1)):
        outs = run(use_graph)
    sync()
    # timed region: exactly K passes, bracketed by barrier + synchronize
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = run(use_graph)
    sync()
    t_rank = time.perf_counter() - t0                                  # this rank's own time (before the closing barrier)
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dev)                       # max over ranks
    # the run's ONE collective on results (RCCL over xGMI when N > 1; SURVEY 8e, vision3d/utils/distributed.py:57-64):
    # all_reduce(SUM) of [sum IR, sum FMR, sum RR, n_pairs, sum t] of the last pass, evaluated on the device harness
    ir, fmr, rr = evaluate(outs)
    gathered = shard.reduce_metrics(shard.metric_vector(ir, fmr, rr, t_rank).to(dev))
    per_rank_s = shard.gather_per_rank(t_rank, dev)
    per_rank_pairs = shard.gather_per_rank(P, dev)
    checksum = torch.zeros(1, dtype=torch.float64) if stub else outs[0]["conf_matrix_pred"].sum().reshape(1).double()
    checksum = shard.gather_metrics(checksum, dev)
    total_pairs = int(gathered["n_pairs"]) * args.steps
    value = total_pairs / elapsed
    out = outs[0]

    result = {
        "metric": METRIC, "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "cpu-stub (control-path test, not a measurement)" if stub else "synthetic",
        "config": {"workload": "cfg2: 3DMatch N=M=%d, C=432, %d denoise steps, max_condition_num=%g (warp active), "
                               "%d independent B=1 pairs per pass per GPU as %d concurrent batch(es), one HIP-graph replay each on its own stream" % (N, S, args.max_condition_num, max(int(p) for p in per_rank_pairs), nstreams),
                   "pairs_per_pass_per_gpu": P, "streams": nstreams, "head_gain": HEAD_GAIN, "denoise_steps": S, "N": N, "M": M, "graph": use_graph,
                   "state": "fp64 (quirk Q2), Sinkhorn arithmetic " + ("fp64 (DR_LOOP_STRICT_F64)" if args.strict_f64 else "fp32"), "gemm_path": "plane images (csrc/pgemm.hip), weights packed once per engine",
                   "gemm_arithmetic": "fp32 in / fp32 out; each product as %s, fp32 accumulate (error vs fp64 = that of an fp32 GEMM)" % SPLIT_TEXT, "parallelism": "pairs sharded over %d GPU(s)" % world},
        "conf_checksum": float(checksum.item()),
        "metric_gather": dict(gathered, collective="all_reduce(SUM) of [sum IR, sum FMR, sum RR, n_pairs, sum t] (float64)",
                              backend="gloo" if stub else ("nccl (RCCL)" if use_dist else "none (1 rank)"),
                              per_rank_pairs=[int(p) for p in per_rank_pairs],
                              per_rank_pairs_per_s=[p * args.steps / t for p, t in zip(per_rank_pairs, per_rank_s)]),
    }

    if rank == 0:
        # (N > 1: rank 0 still measures the roofline of its own GPU after the timed region; single-pair latency, the
        #  Sinkhorn micro-benchmark and the CPU baseline are N = 1 only)
        if world > 1:
            args.no_single_pair = args.no_cpu_baseline = args.no_other_configs = True
        # ---- single-pair latency (the literal "batch=1" of configs[1]) -------------------------------
        if not args.no_single_pair:
            prs1, inp1 = make_inputs(variant, 1, N, M, seed0=7000, device=dev)
            run1 = lambda: eng.run(inp1["f_s"], inp1["f_t"], inp1["p_s"], inp1["p_t"], inp1["x_T"], graph=use_graph)
            for _ in range(3):
                run1()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                run1()
            torch.cuda.synchronize()
            lat = (time.perf_counter() - t1) / 10
            result["single_pair"] = {"ms_per_pair": lat * 1e3, "pairs_per_s": 1.0 / lat}
            # its own roofline: one pair is ~1 400 DEPENDENT launches, and no kernel of that chain -- not even an attention launch
            # that returns at once -- takes less than 4.7 us (profiles/r02_b1_launch_floor.json): the chain alone is above the
            # 5 ms asked for; the FLOP rate beside it shows how far from the arithmetic roof that leaves the pair
            lib.prof_enable(True)
            eng.run(inp1["f_s"], inp1["f_t"], inp1["p_s"], inp1["p_t"], inp1["x_T"], graph=False)
            p1 = lib.prof_collect()
            lib.prof_enable(False)
            n1 = sum(v[0] for v in p1.values())
            fl1 = sum(v[2] for k, v in p1.items() if k in ("gemm", "gemm_split", "attention"))
            result["single_pair"]["roofline"] = {
                "bound": "dependent-launch chain", "launches": n1, "boundary_us": KERNEL_BOUNDARY_US,
                "floor_ms": n1 * KERNEL_BOUNDARY_US * 1e-3, "frac": n1 * KERNEL_BOUNDARY_US * 1e-6 / lat,
                "cheapest_real_kernel_us": DEP_LAUNCH_FLOOR_US, "frac_vs_cheapest_real_kernel": n1 * DEP_LAUNCH_FLOOR_US * 1e-6 / lat,
                "mfma_TFLOPs": fl1 / lat / 1e12, "mfma_frac_of_f32_peak": fl1 / lat / 1e12 / PEAK_MFMA_F32_TFLOPS}

        if not args.no_breakdown:
            # ---- per-kernel-family GPU time of the same pass (eager launches, HIP events on the stream) ---
            run(False)
            torch.cuda.synchronize()
            lib.prof_enable(True)
            reps = 2
            for _ in range(reps):
                run(False)
            prof = lib.prof_collect()
            lib.prof_enable(False)
            tot = sum(v[1] for v in prof.values())
            # (eager launches of ONE of the concurrent batches)
            fam = {k: {"launches_per_pass": v[0] // reps, "ms_per_pass": v[1] / reps, "share": v[1] / tot,
                       "avg_us_per_launch": (v[1] / v[0] * 1e3) if v[0] else 0.0} for k, v in prof.items()}
            dom = max(prof, key=lambda k: prof[k][1])
            c, ms_, work = prof[dom]
            if dom == "gemm_split":
                ach = work / (ms_ * 1e-3) / 1e12
                roof = dict(kernel="%s (family gemm_split)" % SPLIT_KERNEL, bound="mfma", achieved=ach, peak=PEAK_SPLIT_TFLOPS,
                            unit="TFLOP/s", traffic=None, peak_basis="dense 16-bit MFMA peak / %d products per fp32 MAC" % SPLIT_PRODUCTS,
                            peak_f32_mfma=PEAK_MFMA_F32_TFLOPS, frac_vs_f32_mfma_peak=ach / PEAK_MFMA_F32_TFLOPS)
            elif dom in ("gemm", "attention"):
                roof = dict(kernel=dom, bound="mfma", achieved=work / (ms_ * 1e-3) / 1e12, peak=PEAK_MFMA_F32_TFLOPS,
                            unit="TFLOP/s", traffic=None)
            else:
                roof = dict(kernel=dom, bound="hbm", achieved=work / (ms_ * 1e-3) / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s",
                            traffic=None)
            roof["frac"] = roof["achieved"] / roof["peak"]
            if dom == "gemm_split" and os.path.exists(LOOP_PMC):
                # HBM bytes and MFMA-pipe utilisation of THIS kernel over the loop's own launches: separate rocprofv3 --pmc passes of
                # `bench.py --breakdown-only` (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; per launch, averaged like `achieved`)
                tj = json.load(open(LOOP_PMC))
                roof["traffic"] = tj.get("hbm_bytes_per_launch")
                roof["traffic_source"] = "profiles/r06_pgemm_loop_pmc.json (%s; %d launches)" % (tj.get("command"), tj.get("launches_seen", 0))
                roof["mfma_busy_fraction_of_wall_pmc"] = tj.get("derived", {}).get("mfma_busy_fraction_of_wall")
                # what a bare MFMA loop of this kernel's shape sustains on random data under the chip's own clock management (tools/mfma_f16_ceiling.hip):
                # informational -- `peak` stays the nominal figure
                roof["sustained_bare_loop_TFLOPs"] = {"v_mfma_f32_16x16x32_f16 / 3 products": 1941.7 / 3, "source": "profiles/r04_mfma_f16_sustained_ceiling.txt"}
                roof["frac_of_sustained_bare_loop"] = roof["achieved"] / (1941.7 / 3)
                roof["effective_clock_GHz_pmc"] = tj.get("derived", {}).get("effective_clock_GHz")
                roof["avg_us_per_launch_pmc_run"] = tj.get("avg_us_per_launch_profiled")
                if roof["traffic"] and tj.get("avg_us_per_launch_profiled"):
                    # the family's average HBM rate over its own launches: the layer is 125 FLOP per byte, below the balance
                    # point of this ceiling and ~5 TB/s of achievable HBM bandwidth -- the second roof this kernel family sits under
                    roof["hbm_avg_GBps_pmc"] = roof["traffic"] / tj["avg_us_per_launch_profiled"] / 1e3
                    roof["flop_per_hbm_byte"] = (work / c) / roof["traffic"]
            roof["avg_us_per_launch"] = ms_ / c * 1e3
            roof["work_per_launch"] = work / c
            roof["note"] = ("dominant family by GPU time; achieved = algorithmic fp32 FLOPs (2*rows*cols*K per GEMM) / time; "
                            "the kernel computes each fp32 product as " + SPLIT_TEXT + " accumulated in fp32 "
                            "(fp32-level accuracy: tests/test_planes_gpu.py holds every op of the chain to float64 products)")
            roof["measured_on"] = "eager launches of one batch of %d pairs (HIP events on the launch stream)" % per[0]
            if dom == "gemm_split" and not args.breakdown_only:
                try:
                    po = pgemm_per_op(dev, per[0] * (N + M))
                    roof["per_op"] = po["ops"]
                    roof["per_op_measured"] = po["measured"] + "; rows = %d" % po["rows"]
                except Exception as e:                  # (a diagnostic block must not take the line down)
                    roof["per_op_error"] = "%s: %s" % (type(e).__name__, e)
            result["roofline"] = roof
            result["kernel_families"] = fam
        if sk_roof is not None:
            result["sinkhorn_roofline"] = sk_roof
        if "roofline" not in result and "sinkhorn_roofline" in result:
            result["roofline"] = result["sinkhorn_roofline"]
        if not args.no_other_configs:
            del outs, out
            eng._graphs.clear()                     # the headline's 2 x 128-pair static buffers and graphs are not needed any more
            torch.cuda.empty_cache()
            result["other_configs"] = other_configs(dev)
        if not args.no_cpu_baseline:
            result["cpu_baseline"], kept = cpu_baseline(variant, N, M, S, args.max_condition_num)
            if kept:
                result["ir_fmr_parity"] = ir_fmr_parity(eng, variant, N, M, kept, dev, HEAD_GAIN)
                try:
                    result["ir_fmr_parity"]["stress_head"] = stress_head_parity(variant, N, M, S, args.max_condition_num, [k["seed"] for k in kept], dev)
                except Exception as e:                  # a secondary block must not take the line down -- but it must show: parity_ok below goes false
                    result["ir_fmr_parity"]["stress_head"] = {"error": "%s: %s" % (type(e).__name__, e)}
                par, sh = result["ir_fmr_parity"], result["ir_fmr_parity"]["stress_head"]
                # top-level flag: the primary sample entirely inside the 1e-4 contract, IR / FMR of BOTH heads within north_star's 0.1, no block errored
                result["parity_ok"] = bool(par["pairs_within_1e4"] == par["pairs"] and par["max_abs_ir_diff"] <= 0.1 and abs(par["fmr_hip"] - par["fmr_oracle"]) <= 0.1
                                           and "error" not in sh and sh["max_abs_ir_diff"] <= 0.1 and abs(sh["fmr_hip"] - sh["fmr_oracle"]) <= 0.1)
    if rank == 0:
        details = write_details(result)
        line = json.dumps(compact_line(result, details), separators=(", ", ": "))
        # the driver keeps a bounded tail of stdout (the r05 line of 25.5 KB did not survive it): the line is a SUMMARY, everything else is in the file
        assert len(line) < MAX_LINE_BYTES, "bench.py: the JSON line is %d bytes (limit %d): move the new block to bench_details.json" % (len(line), MAX_LINE_BYTES)
        sys.stdout.flush()
        print(line, flush=True)
    if use_dist:
        dist.barrier()                      # the other ranks wait for rank 0's post-measurements before tearing down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
