"""Per-step, TEACHER-FORCED parity of the reverse-diffusion loops at the two configurations bench.py times with the warp active:
cfg2 (3DMatch 256 x 256, 20 steps, 128 pairs per call: the plane path, exactly the headline's batch shape) and cfg5 (2D-3D
1024 x 2048, 10 steps, 8 pairs per call: plane path, multi-launch Sinkhorn, chip-wide top-K).  Needs a GPU.

Why teacher forcing.  The loop feeds a top-K selection back into itself (pipeline.py:293-309: K largest of N M confidences -> weighted
Kabsch -> warp -> denoiser input).  A selection is discontinuous: where the K-th and (K+1)-th confidence are equal to the last bits, ANY
two float32 evaluations (the reference on another BLAS included) may select different sets, the fits differ, and the two trajectories
part for good -- a free-running comparison can then say nothing about the steps behind that one (tests/test_2d3d_gpu.py stops at the
first such step; bench.py's parity sample shows four such pairs).  Here every step k is an independent evaluation on the device of ONE
pass through the loop body (3D/models/pipeline.py:237-256 / EXP/model.py:637-680) from the state the ORACLE's run had on entering step k
(dr_loop_trace.force_x), warped with the ORACLE's pose of that step (dr_loop_trace.force_R / force_t), through dr_denoise_loop /
dr_denoise_loop_2d3d themselves -- the same kernels, batch shape and launch sequence as the timed loop.  All steps are asserted, no break:

  warp confidences (Sinkhorn of the forced state)           |hip - oracle| <= 1e-4 on every entry (measured ~1e-8)
  top-K selection (index work)                              equal to the oracle's as a set, or differing ONLY inside the tied band around
                                                            the oracle's K-th value (band = what the confidences' own deviation can flip)
  the fit (R_forwd, t_forwd, cond)                          1e-4 against the oracle's weighted Kabsch on the set the DEVICE selected (always), and
                                                            against the oracle's own pose wherever the two sets are equal
  x_start (denoiser + matching head on the oracle's warp)   plain 1e-4 on every entry (soft head, 2D-3D); where that fails on the stress head (HEAD_GAIN 24,
                                                            logits in the thousands) the exemption rule of tests/test_loop_gpu.py against a float64
                                                            evaluation of that one step, entry-wise or -- see stress_tile_parity -- tile-wise
  the DDIM update                                           the device's next state against the reference's update arithmetic applied to the
                                                            device's own x_start (1e-9), and against the oracle's next state (1e-4 where x_start holds)
"""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, weights, masks
from tests.test_loop_gpu import engine, assert_matrix_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _bounded_host_threads():
    n = torch.get_num_threads()
    torch.set_num_threads(min(16, n))          # (the oracle is ~1 500 small torch ops per pair: more threads only add synchronisation)
    yield
    torch.set_num_threads(n)


class OracleSelection:
    """the oracle's top-K of one warp-confidence tile (procrustes.py:66-70: the K largest of N M values), computed once per (scene, step)"""
    def __init__(self, conf_orc, K):
        self.conf, self.K = conf_orc, K
        part = np.argpartition(-conf_orc, K)[:K + 1]              # the K + 1 largest, unordered
        part = part[np.argsort(-conf_orc[part], kind="stable")]
        self.set = set(part[:K].tolist())
        self.vK, self.vK1 = float(conf_orc[part[K - 1]]), float(conf_orc[part[K]])
        self.gap_rel = (self.vK - self.vK1) / self.vK


def selection_check(sel, idx_hip, conf_dev):
    """sel: OracleSelection; idx_hip: the device's selected flat indices; conf_dev: the device's confidences of the tile.
    -> sets_equal.  Asserts that an entry of the symmetric difference lies inside the tied band around the oracle's K-th value: the band is
    what the deviation between the two evaluations of the confidences can flip (4 x their largest difference, at least 3e-6 relative)."""
    s_hip = set(int(i) for i in idx_hip.tolist() if i >= 0)
    assert len(s_hip) == sel.K, ("the device selected %d entries, K = %d" % (len(s_hip), sel.K))
    if s_hip == sel.set:
        return True
    band = max(4.0 * float(np.abs(conf_dev - sel.conf).max()), 3e-6 * sel.vK)
    diff = np.fromiter(s_hip ^ sel.set, dtype=np.int64)
    off = np.abs(sel.conf[diff] - sel.vK)
    assert off.max() <= band, ("selections differ OUTSIDE the tied band", float(off.max()), band, sel.vK, len(diff))
    return False


def kabsch_on(conf_orc, idx, p_s, p_t, M):
    """the reference's weighted fit (procrustes.py:17-44, restated in oracle.kabsch) on a given selection with the ORACLE's confidences"""
    idx = torch.as_tensor(np.asarray([i for i in idx.tolist() if i >= 0]), dtype=torch.int64)
    w = torch.from_numpy(conf_orc)[idx].float()
    return orc.kabsch(p_s[0][idx // M][None], p_t[0][idx % M][None], w[None, :, None])


def ddim_expected(x_in, x0, t, tn, shift_min, first):
    """pipeline.py:246-256 in the reference's own dtype choreography (oracle.denoise_loop) on a given x_start"""
    ac, sra, srm1 = orc.diffusion_schedule()
    x = x_in
    if shift_min:
        x = x - x.min()
    eps = (sra[t].view(1, 1) * x - x0) / srm1[t].view(1, 1)
    sigma, c, sqrt_an = orc.ddim_coefficients(ac, t, tn)
    return x0 * sqrt_an + c * eps


# stress_tile_parity: how much further from float64 than the reference's own float32 run the device may be on the STRESS head's worst tile.
# Measured over the 20 steps x 128 slots of the bench seeds (MI355X): 832 of the 2 560 step evaluations miss the plain 1e-4 somewhere,
# 512 of those pass the entry-wise exemption rule, 320 need the tile rule; there the device is at most 2.5e-4 from float64 and at most 6.6 x the
# reference's distance (6 entries per tile beyond 1e-4 of the reference at worst).
# What this is NOT (round 6, profiles/r06_stress_head_tile_rule_cause.json): an artefact of the plane path's 22-bit operands.  With the head's
# projection on 24-bit operands (DR_HEAD_F32=1) 304 tiles still need the rule; with EVERY layer GEMM and the attention on the f32-input MFMA
# kernels (exact fp32 products, DR_PLANES=0) 407 tiles need it, at the same worst distances (2.4e-4, 4.5 x).  At matching logits in the
# thousands one float32 ulp of a logit is ~1e-4 of x_start: any two correct fp32 evaluations of the denoiser (another summation order, another
# BLAS) differ by a few 1e-4 on the sharp entries.  The same kernels hold the SOFT head (logits O(10)) to 5.5e-6 on every entry of every step.
TILE_FACTOR = 8.0
TILE_ABS_CAP = 5e-4


def stress_tile_parity(got, ref, f64, what, stats):
    """x_start tiles of the STRESS head that miss the plain 1e-4 (matching logits in the thousands: one float32 ulp of a logit is ~1e-4 of
    x_start, so any two float32 evaluations of the head -- the reference on another BLAS included -- differ by that much on the sharp entries).
    First the entry-wise exemption rule of tests/test_loop_gpu.py (an entry is exempt where the reference's OWN float32 value is > 2e-5 from the
    float64 evaluation of this step; exempt entries must be as close to float64 as twice the reference).  That rule samples ONE realisation of
    the reference's rounding: an entry where the reference happened to round well is held to 1e-4 although it is as ill-conditioned as its
    neighbours.  Where it fails, a tile-level statement decides: the device's LARGEST distance from float64 over the tile is at most
    TILE_FACTOR x the reference's largest and at most TILE_ABS_CAP (see the constants' comment for what was measured and why the factor is
    not 2), and the entries beyond 1e-4 of the reference stay as few as an exemption list may be long."""
    try:
        assert_matrix_parity(got, ref, f64, what)
        return
    except AssertionError:
        pass
    got, ref, f64 = (np.asarray(a, dtype=np.float64).ravel() for a in (got, ref, f64))
    e_hip, e_ref = float(np.abs(got - f64).max()), float(np.abs(ref - f64).max())
    n_far = int((np.abs(got - ref) > 1e-4).sum())
    stats["tile_rule_used"] = stats.get("tile_rule_used", 0) + 1
    stats["tile_rule_worst_hip_over_ref"] = max(stats.get("tile_rule_worst_hip_over_ref", 0.0), e_hip / max(e_ref, 1e-12))
    stats["tile_rule_worst_e_hip"] = max(stats.get("tile_rule_worst_e_hip", 0.0), e_hip)
    stats["tile_rule_worst_n_far"] = max(stats.get("tile_rule_worst_n_far", 0), n_far)
    assert e_hip <= max(1e-4, TILE_FACTOR * e_ref) and e_hip <= TILE_ABS_CAP, (what, "further from float64 than TILE_FACTOR x the reference", e_hip, e_ref)
    assert n_far <= 0.005 * got.size + 40, (what, "too many entries beyond 1e-4", n_far)


# -----------------------------------------------------------------------------------------------------------------------------------
# cfg2: the bench's parity seeds (1001 .. 1010; 1005 / 1006 / 1007 / 1010 are the ones whose free-running trajectories part on the
# stress head), tiled over the 128 slots of one plane-path call
# -----------------------------------------------------------------------------------------------------------------------------------
BENCH_SEEDS = list(range(1001, 1011))


@pytest.mark.parametrize("family", ["main", "soft"])
def test_cfg2_every_step_teacher_forced(family):
    variant, N, M, steps, mc, P = "3dmatch", 256, 256, 20, 200.0, 128
    v = synth.VARIANTS[variant]
    W = weights(variant, family)
    ms, mt = masks(N, M)
    prs, trs = [], []
    for sd in BENCH_SEEDS:
        p = synth.make_pair(N, M, v["C"], seed=sd)
        q = {k: T(p[k])[None] for k in ("src_feats", "tgt_feats", "s_pcd", "t_pcd", "x_T")}
        tr = []
        orc.denoise_loop(W, v, q["src_feats"], q["tgt_feats"], q["s_pcd"], q["t_pcd"], ms, mt, q["x_T"], steps, mc, variant=variant, trace=tr)
        prs.append(q); trs.append(tr)
    nd = len(BENCH_SEEDS)
    slot = [i % nd for i in range(P)]
    cat = lambda k: torch.cat([prs[s][k] for s in slot]).to(DEV)
    # the state entering step k: x_T, then the oracle's state after step k - 1 (float64 from step 1 on, quirk Q2)
    fx = torch.stack([torch.cat([(prs[s]["x_T"].double() if k == 0 else trs[s][k - 1]["x"].double()) for s in slot]) for k in range(steps)])
    fR = torch.stack([torch.cat([trs[s][k]["R_forwd"] for s in slot]) for k in range(steps)])
    ft = torch.stack([torch.cat([trs[s][k]["t_forwd"] for s in slot]) for k in range(steps)])
    eng = engine(variant, steps, mc, family)               # 128 pairs = 65 536 token rows: the size rule takes the plane path
    out = eng.run(cat("src_feats"), cat("tgt_feats"), cat("s_pcd"), cat("t_pcd"), cat("x_T"), trace="full", force=dict(x=fx, R=fR, t=ft))
    torch.cuda.synchronize()
    del fx
    K = int(max(N, M) * v["sample_rate"])
    times = orc.time_pairs(steps)
    W64 = None
    stats = dict(sets_differ=0, steps_with_exemptions=0, worst_x0=0.0, worst_R=0.0, worst_wconf=0.0, min_gap_rel=1.0)
    for k in range(steps):
        sels = [OracleSelection(trs[s][k]["conf"][0].numpy().reshape(-1), K) for s in range(nd)]
        stats["min_gap_rel"] = min([stats["min_gap_rel"]] + [z.gap_rel for z in sels])
        wconf = out["wconf"][k].cpu().numpy().reshape(P, -1)
        topk = out["topk_idx"][k].cpu().numpy()
        Rk, tk, ck = out["R_forwd"][k].cpu(), out["t_forwd"][k].cpu(), out["cond"][k].cpu()
        x0k, xnk = out["x0"][k].cpu(), out["x_next"][k].cpu()
        for pi in range(P):
            s = slot[pi]
            rec, q = trs[s][k], prs[s]
            c_orc = sels[s].conf
            # -- Sinkhorn of the forced state (min-shift fused, float64 state in, float32 out)
            dw = float(np.abs(wconf[pi] - c_orc).max())
            assert dw <= 1e-4, (k, pi, dw)
            stats["worst_wconf"] = max(stats["worst_wconf"], dw)
            # -- the selection, then the fit on the selection the device made
            same = selection_check(sels[s], topk[pi], wconf[pi])
            R_sel, t_sel, cond_sel = kabsch_on(c_orc, topk[pi], q["s_pcd"], q["t_pcd"], M)
            ok = bool(cond_sel[0] < mc)
            R_exp = R_sel[0] if ok else torch.eye(3)
            t_exp = t_sel[0] if ok else torch.zeros(3, 1)
            eR = float((Rk[pi] - R_exp).abs().max()); et = float((tk[pi] - t_exp).abs().max())
            assert eR < 1e-4 and et < 1e-4, (k, pi, eR, et)
            assert abs(float(ck[pi]) - float(cond_sel[0])) <= 1e-4 * float(cond_sel[0]), (k, pi, float(ck[pi]), float(cond_sel[0]))
            stats["worst_R"] = max(stats["worst_R"], eR)
            if same:
                assert float((Rk[pi] - rec["R_forwd"][0]).abs().max()) < 1e-4 and float((tk[pi] - rec["t_forwd"][0]).abs().max()) < 1e-4, (k, pi)
            else:
                stats["sets_differ"] += 1
            # -- x_start on the oracle's warp: plain bound first, the exemption rule (float64 evaluation of this one step) where it fails
            x0_ref = rec["x0"][0]
            d = float((x0k[pi] - x0_ref).abs().max())
            stats["worst_x0"] = max(stats["worst_x0"], d)
            if d > 1e-4:
                assert family == "main", ("the soft family holds a plain 1e-4", k, pi, d)
                if W64 is None:
                    W64 = {kk: t_.double() for kk, t_ in W.items()}
                hs, ht, pe_s, pe_t = orc.denoiser(W64, v, q["src_feats"].double(), q["tgt_feats"].double(), rec["warped"], q["t_pcd"], ms, mt)
                x0_64 = orc.match_head(W64, v, hs, ht, pe_s, pe_t, ms, mt)[0].numpy()
                stress_tile_parity(x0k[pi].numpy(), x0_ref.numpy(), x0_64, "x_start, step %d slot %d" % (k, pi), stats)
                stats["steps_with_exemptions"] += 1
            # -- the update: the reference's arithmetic on the device's own x_start, and the oracle's next state
            x_in = q["x_T"][0] if k == 0 else trs[s][k - 1]["x"][0]
            exp = ddim_expected(x_in, x0k[pi], times[k][0], times[k][1], True, k == 0)
            assert float((xnk[pi] - exp.double()).abs().max()) <= 1e-9, (k, pi)
            if d <= 1e-4:
                assert float((xnk[pi] - rec["x"][0].double()).abs().max()) <= 1e-4, (k, pi)
    print("cfg2 teacher-forced (%s): %s" % (family, stats))


# -----------------------------------------------------------------------------------------------------------------------------------
# cfg5: 8 pairs per call (24 576 token rows: plane path, grid-form Sinkhorn, chip-wide top-K), max_condition_num = 200, padding masks and
# a different tgt_mask_da -- what bench.py's other_configs.cfg5.P8 times.  ONE 10-step call: slot p carries scene p % 2, all ten loop
# steps are forced, so loop step k of slot p is step k of scene p % 2 (two scenes with different masks, each in four slots of the batch).
# -----------------------------------------------------------------------------------------------------------------------------------
def test_cfg5_every_step_teacher_forced_batch8():
    from diffreg_hip.engine import DenoiseEngine2D3D
    N, M, steps, mc, P = 1024, 2048, 10, 200.0, 8
    cfgv = synth.VARIANTS["2d3d"]
    Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
    W = {k: T(a) for k, a in Wn.items()}
    seeds = [51, 52]
    nv, mv, mda = [1000, 1024], [2000, 2048], [1900, 2000]
    keys = ("img_feats", "img_dino", "img_pixels", "pcd_feats", "s_pcd", "t_pcd_da", "x_T")
    prs, trs, mks = [], [], []
    for i, sd in enumerate(seeds):
        pr = synth.make_pair_2d3d(N, M, sd, weights=Wn)
        q = {k: T(pr[k])[None] for k in keys}
        ms, mt = masks(N, M, nv[i], mv[i])
        mt_da = torch.arange(M)[None] < mda[i]
        tr = []
        orc.denoise_loop_2d3d(W, cfgv, q["img_feats"], q["img_dino"], q["img_pixels"], q["pcd_feats"], q["s_pcd"], q["t_pcd_da"], ms, mt, mt_da,
                              q["x_T"], steps, mc, trace=tr)
        prs.append(q); trs.append(tr); mks.append((ms, mt, mt_da))
    slot = [i % len(seeds) for i in range(P)]
    cat = lambda k: torch.cat([prs[s][k] for s in slot]).to(DEV)
    dm = tuple(torch.cat([mks[s][j] for s in slot]).to(DEV) for j in range(3))
    fx = torch.stack([torch.cat([(prs[s]["x_T"].double() if k == 0 else trs[s][k - 1]["x"].double()) for s in slot]) for k in range(steps)])
    fR = torch.stack([torch.cat([trs[s][k]["R_forwd"] for s in slot]) for k in range(steps)])
    ft = torch.stack([torch.cat([trs[s][k]["t_forwd"] for s in slot]) for k in range(steps)])
    eng = DenoiseEngine2D3D(W, steps=steps, max_condition_num=mc, device=DEV)        # planes=None: the size rule (24 576 rows -> plane path)
    out = eng.run(*[cat(k) for k in keys], masks=dm, trace="full", force=dict(x=fx, R=fR, t=ft))
    torch.cuda.synchronize()
    times = orc.time_pairs(steps)
    bin_score = W["denoising_coarse_matching.bin_score"]
    stats = dict(sets_differ=0, worst_x0=0.0, worst_R=0.0, worst_wconf=0.0, min_gap_rel=1.0)
    for k in range(steps):
        wconf = out["wconf"][k].cpu().numpy().reshape(P, -1)
        topk = out["topk_idx"][k].cpu().numpy()
        Rk, tk, ck = out["R_forwd"][k].cpu(), out["t_forwd"][k].cpu(), out["cond"][k].cpu()
        x0k, xnk = out["x0"][k].cpu(), out["x_next"][k].cpu()
        sels, xms = [], []
        for s in range(len(seeds)):
            # the oracle's warp confidences of this step (its trace keeps the pose only): Sinkhorn of the state that entered the step
            ms, mt, mt_da = mks[s]
            xm = (prs[s]["x_T"] if k == 0 else trs[s][k - 1]["x"]).clone().masked_fill_(~orc.pair_mask(ms, mt_da), float("-inf"))
            c_orc = orc.sinkhorn_log(xm, bin_score, 3, ms, mt_da).exp()[0, :-1, :-1].float().numpy().reshape(-1)
            Kp = int(max(int(ms.sum()), int(mt_da.sum())) * cfgv["sample_rate"])
            sels.append(OracleSelection(c_orc, Kp)); xms.append(xm)
            stats["min_gap_rel"] = min(stats["min_gap_rel"], sels[-1].gap_rel)
        for pi in range(P):
            s = slot[pi]
            rec, q, (ms, mt, mt_da) = trs[s][k], prs[s], mks[s]
            c_orc, xm = sels[s].conf, xms[s]
            dw = float(np.abs(wconf[pi] - c_orc).max())
            assert dw <= 1e-4, (k, pi, dw)
            stats["worst_wconf"] = max(stats["worst_wconf"], dw)
            same = selection_check(sels[s], topk[pi], wconf[pi])
            R_sel, t_sel, cond_sel = kabsch_on(c_orc, topk[pi], q["s_pcd"], q["t_pcd_da"], M)
            ok = bool(cond_sel[0] < mc)
            R_exp = R_sel[0] if ok else torch.eye(3)
            t_exp = t_sel[0] if ok else torch.zeros(3, 1)
            eR = float((Rk[pi] - R_exp).abs().max()); et = float((tk[pi] - t_exp).abs().max())
            assert eR < 1e-4 and et < 1e-4, (k, pi, eR, et)
            assert abs(float(ck[pi]) - float(cond_sel[0])) <= 1e-4 * float(cond_sel[0]), (k, pi, float(ck[pi]), float(cond_sel[0]))
            stats["worst_R"] = max(stats["worst_R"], eR)
            if same:
                assert float((Rk[pi] - rec["R_forwd"][0]).abs().max()) < 1e-4 and float((tk[pi] - rec["t_forwd"][0]).abs().max()) < 1e-4, (k, pi)
            else:
                stats["sets_differ"] += 1
            # x_start on the oracle's warp: plain 1e-4 on every entry (the 2D-3D exemption lists are empty by measurement)
            d = float((x0k[pi] - rec["x0"][0]).abs().max())
            assert d <= 1e-4, (k, pi, d)
            stats["worst_x0"] = max(stats["worst_x0"], d)
            # the update: masked entries stay -inf (quirk Q8), the rest follows the reference's arithmetic on the device's x_start
            valid = orc.pair_mask(ms, mt_da)[0]
            exp = ddim_expected(xm[0], x0k[pi], times[k][0], times[k][1], False, k == 0).double()
            got = xnk[pi]
            assert torch.equal(torch.isfinite(got), valid)
            assert float((got[valid] - exp[valid]).abs().max()) <= 1e-9, (k, pi)
            assert float((got[valid] - rec["x"][0].double()[valid]).abs().max()) <= 1e-4, (k, pi)
    print("cfg5 teacher-forced, 8 pairs per call:", stats)
