"""GPU parity of the evaluation harness through the C ABI: dr_inlier_ratio_f32, dr_nrfmr_f32,
dr_ransac_corr_f64, dr_registration_recall_f64 against oracle/metrics_oracle.py and the reference-minted vectors."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as mo
from tests.helpers import metrics_scene

pytestmark = pytest.mark.gpu
CASES = [(256, 256, 3), (200, 256, 5), (512, 512, 8)]


def dev(x):
    return x.cuda() if torch.is_tensor(x) else x


def seg_of(sc, cap=None):
    from diffreg_hip.metrics import segment_matches
    seg, cnt = segment_matches(sc["matches"].cuda(), 1)
    if cap is not None:          # the loop's own layout: cap = N + M, rows beyond count are garbage
        pad = torch.full((1, cap, 3), 7, dtype=torch.int64, device="cuda")
        pad[:, :seg.shape[1]] = seg
        seg = pad
    return seg, cnt


@pytest.mark.parametrize("N,M,seed", CASES)
def test_inlier_ratio_vs_reference_vectors(golden, N, M, seed):
    from diffreg_hip import lib
    g = golden("metrics_ref")
    tag = "%dx%d_s%d_" % (N, M, seed)
    sc = metrics_scene(N, M, seed)
    seg, cnt = seg_of(sc, cap=N + M)
    ir, n = lib.inlier_ratio(seg, cnt, dev(sc["s_pcd"]), dev(sc["t_pcd"]), dev(sc["rot"]), dev(sc["trn"]), 0.1)
    assert np.array_equal(ir.cpu().numpy(), g[tag + "ir3d"])          # bit-exact: integer counts, one float32 division
    ir4, _ = lib.inlier_ratio(seg, cnt, dev(sc["s_pcd"]), dev(sc["t_pcd4"]), dev(sc["rot"]), dev(sc["trn"]), 0.04,
                              dev(sc["coarse_flow"])[None])
    assert np.array_equal(ir4.cpu().numpy(), g[tag + "ir4d"])
    # without the flow the non-rigid targets are missed: the flow argument is really used
    ir0, _ = lib.inlier_ratio(seg, cnt, dev(sc["s_pcd"]), dev(sc["t_pcd4"]), dev(sc["rot"]), dev(sc["trn"]), 0.04)
    assert float(ir0) < float(ir4)


def test_inlier_ratio_batched_and_edge_cases():
    from diffreg_hip import lib
    scs = [metrics_scene(200, 256, s) for s in (5, 6, 7)]
    P, cap = 5, 700
    matches = torch.zeros(P, cap, 3, dtype=torch.int64)
    count = torch.zeros(P, dtype=torch.int32)
    for p, sc in enumerate(scs):
        k = len(sc["matches"])
        matches[p, :k] = sc["matches"]
        count[p] = k
    matches[3, :2] = scs[0]["matches"][:2]; count[3] = 2            # fewer than 3 matches -> 0 (loss.py:403-404)
    count[4] = 0                                                     # no matches at all
    stack = lambda key: torch.cat([scs[min(p, 2)][key] for p in range(P)])
    ir, n = lib.inlier_ratio(matches.cuda(), count.cuda(), stack("s_pcd").cuda(), stack("t_pcd").cuda(), stack("rot").cuda(),
                             stack("trn").cuda(), 0.1)
    for p, sc in enumerate(scs):
        want = mo.inlier_ratio(sc["matches"], sc["s_pcd"], sc["t_pcd"], sc["rot"], sc["trn"], 0.1)
        assert float(ir[p]) == float(want[0])
    assert float(ir[3]) == 0.0 and float(ir[4]) == 0.0 and int(n[4]) == 0
    with pytest.raises(RuntimeError):
        lib.inlier_ratio(matches, count, stack("s_pcd"), stack("t_pcd"), stack("rot"), stack("trn"), 0.1)   # CPU tensors


def test_segment_matches_accepts_any_row_order():
    """MatchMetrics takes the flat [K, 3] (b, i, j) list in ANY row order, like the reference's compute_inlier_ratio
    (an unsorted list used to scatter to wrong slots)"""
    from diffreg_hip.metrics import segment_matches
    g = torch.Generator().manual_seed(2)
    B, K = 4, 37
    mp = torch.stack([torch.randint(0, B, (K,), generator=g), torch.randint(0, 50, (K,), generator=g), torch.randint(0, 60, (K,), generator=g)], 1)
    mp[mp[:, 0] == 2, 0] = 3                                          # one pair without matches
    seg, cnt = segment_matches(mp.cuda(), B)
    for b in range(B):
        rows = mp[mp[:, 0] == b]                                       # stable: the original relative order inside a pair
        assert int(cnt[b]) == len(rows)
        assert torch.equal(seg[b, :len(rows)].cpu(), rows)


@pytest.mark.parametrize("N,M,seed", CASES)
def test_nrfmr_vs_reference_vectors(golden, N, M, seed):
    from diffreg_hip import lib
    g = golden("metrics_ref")
    tag = "%dx%d_s%d_" % (N, M, seed)
    sc = metrics_scene(N, M, seed)
    seg, cnt = seg_of(sc, cap=N + M)
    Q, Rn = len(sc["metric_index"]), len(sc["raw_pcd"])
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
    r, n, bl = lib.nrfmr(seg, cnt, dev(sc["s_pcd"]), dev(sc["t_pcd4"]), dev(sc["raw_pcd"]), dev(sc["raw_flow"]), i32([0, Rn]),
                         dev(sc["metric_index"]), i32([0, Q]), Q, dev(sc["rot"]), dev(sc["trn"]), 0.1, 0.04, want_blended=True)
    # float32 arithmetic in the reference's order.  Matches that share a source point are anchors at the same place; which
    # of them np.argpartition keeps is unspecified upstream, the kernel takes the lowest match row (oracle: stable=True)
    want, per, blends = mo.nrfmr(sc["matches"], sc["s_pcd"], sc["t_pcd4"], [sc["raw_pcd"]], [sc["raw_flow"]], [sc["metric_index"]],
                                 sc["rot"], sc["trn"], 0.04, stable=True)
    assert np.abs(bl.cpu().numpy() - blends[0]).max() <= 2e-7
    assert abs(float(r) - float(want)) <= 1.0 / Q + 1e-7 and int(n) == round(float(r) * Q)
    # against the reference's own output: identical wherever no tie reaches the third neighbour
    m = sc["matches"]
    d = np.sort(np.sum((sc["s_pcd"][0][m[:, 1]].numpy()[None] - sc["raw_pcd"][sc["metric_index"]].numpy()[:, None]) ** 2, -1), 1)
    untied = (d[:, 0] != d[:, 1]) & (d[:, 1] != d[:, 2]) & (d[:, 2] != d[:, 3])
    assert untied.sum() > 100 and np.abs(bl.cpu().numpy() - g[tag + "blended"])[untied].max() <= 2e-7
    assert abs(float(r) - float(g[tag + "nrfmr"])) <= (Q - untied.sum() + 1.0) / Q


def test_nrfmr_two_pairs_and_too_few_anchors():
    from diffreg_hip import lib
    from diffreg_hip.metrics import compute_nrfmr
    a, b = metrics_scene(200, 256, 5), metrics_scene(200, 256, 6)
    mb = b["matches"].clone(); mb[:, 0] = 1
    mp = torch.cat([a["matches"], mb]).cuda()
    data = dict(s_pcd=torch.cat([a["s_pcd"], b["s_pcd"]]).cuda(), t_pcd=torch.cat([a["t_pcd4"], b["t_pcd4"]]).cuda(),
                src_pcd_list=[a["raw_pcd"].cuda(), b["raw_pcd"][:1500].cuda()], sflow_list=[a["raw_flow"].cuda(), b["raw_flow"][:1500].cuda()],
                metric_index_list=[a["metric_index"].cuda(), b["metric_index"][b["metric_index"] < 1500].cuda()],
                batched_rot=torch.cat([a["rot"], b["rot"]]).cuda(), batched_trn=torch.cat([a["trn"], b["trn"]]).cuda())
    got = float(compute_nrfmr(mp, data, 0.04))
    cpu = {k: ([x.cpu() for x in v] if isinstance(v, list) else v.cpu()) for k, v in data.items()}
    want, per, _ = mo.nrfmr(mp.cpu(), cpu["s_pcd"], cpu["t_pcd"], cpu["src_pcd_list"], cpu["sflow_list"], cpu["metric_index_list"],
                            cpu["batched_rot"], cpu["batched_trn"], 0.04, stable=True)
    assert abs(got - float(want)) <= 2e-3
    # fewer than 4 anchors: the reference's argpartition raises; the kernel reports 0 for that pair
    seg, cnt = seg_of(a)
    cnt[:] = 3
    Q = len(a["metric_index"])
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
    r, n = lib.nrfmr(seg, cnt, dev(a["s_pcd"]), dev(a["t_pcd4"]), dev(a["raw_pcd"]), dev(a["raw_flow"]), i32([0, len(a["raw_pcd"])]),
                     dev(a["metric_index"]), i32([0, Q]), Q, dev(a["rot"]), dev(a["trn"]))
    assert float(r) == 0.0 and int(n) == 0


@pytest.mark.parametrize("N,M,seed,iters", [(256, 256, 3, 4000), (200, 256, 5, 3000), (512, 512, 8, 50000)])
def test_ransac_vs_oracle(N, M, seed, iters):
    from diffreg_hip import lib
    sc = metrics_scene(N, M, seed)
    seg, cnt = seg_of(sc, cap=N + M)
    r = lib.ransac_corr(seg, cnt, dev(sc["s_pcd"]), dev(sc["t_pcd"]), 0.05, iters, seed=11, pair_ids=torch.tensor([4]))
    o = mo.ransac_corr(sc["s_pcd"][0].numpy(), sc["t_pcd"][0].numpy(), sc["matches"][:, 1:].numpy(), 0.05, iters, seed=11, pair_id=4)
    K = len(sc["matches"])
    assert round(float(r["fitness"]) * K) == o["n_inlier"]                       # same best score (integer)
    assert float(r["inlier_rmse"]) == pytest.approx(o["inlier_rmse"], rel=1e-9)
    assert np.abs(r["rot"][0].cpu().numpy() - o["R"]).max() < 1e-9              # same hypothesis (or a permutation of its triple)
    assert np.abs(r["trn"][0, :, 0].cpu().numpy() - o["t"]).max() < 1e-9
    assert np.abs(o["R"] - sc["pair"]["R_gt"]).max() < 0.02
    # deterministic: the same call gives the same bits
    r2 = lib.ransac_corr(seg, cnt, dev(sc["s_pcd"]), dev(sc["t_pcd"]), 0.05, iters, seed=11, pair_ids=torch.tensor([4]))
    assert torch.equal(r["rot"], r2["rot"]) and torch.equal(r["best_iter"], r2["best_iter"])


def test_ransac_batched_degenerate_and_sharding():
    from diffreg_hip import lib
    scs = [metrics_scene(200, 256, s) for s in (5, 6)]
    P, cap = 4, 456
    matches = torch.zeros(P, cap, 3, dtype=torch.int64)
    count = torch.zeros(P, dtype=torch.int32)
    for p, sc in enumerate(scs):
        matches[p, :len(sc["matches"])] = sc["matches"]; count[p] = len(sc["matches"])
    matches[2, :2] = scs[0]["matches"][:2]; count[2] = 2                         # < 3 matches: identity (loss.py:363-366)
    matches[3, :10, 1] = 0; matches[3, :10, 2] = torch.arange(10); count[3] = 10  # every triple repeats a source point
    stack = lambda key: torch.cat([scs[min(p, 1)][key] for p in range(P)]).cuda()
    ids = torch.tensor([10, 11, 12, 13])
    r = lib.ransac_corr(matches.cuda(), count.cuda(), stack("s_pcd"), stack("t_pcd"), 0.05, 3000, seed=2, pair_ids=ids)
    eye = torch.eye(3, dtype=torch.float64, device="cuda")
    for p in (2, 3):
        assert torch.equal(r["rot"][p], eye) and float(r["trn"][p].abs().sum()) == 0 and int(r["best_iter"][p]) == -1
    for p, sc in enumerate(scs):
        o = mo.ransac_corr(sc["s_pcd"][0].numpy(), sc["t_pcd"][0].numpy(), sc["matches"][:, 1:].numpy(), 0.05, 3000, seed=2, pair_id=10 + p)
        assert np.abs(r["rot"][p].cpu().numpy() - o["R"]).max() < 1e-9
    # a shard that holds only pair 11 draws the same hypotheses for it
    r1 = lib.ransac_corr(matches[1:2].cuda(), count[1:2].cuda(), stack("s_pcd")[1:2], stack("t_pcd")[1:2], 0.05, 3000, seed=2,
                         pair_ids=ids[1:2])
    assert torch.equal(r1["rot"][0], r["rot"][1]) and torch.equal(r1["best_iter"][0], r["best_iter"][1])


@pytest.mark.parametrize("N,M,seed", CASES)
def test_registration_recall_vs_reference_vectors(golden, N, M, seed):
    from diffreg_hip import lib
    g = golden("metrics_ref")
    tag = "%dx%d_s%d_" % (N, M, seed)
    sc = metrics_scene(N, M, seed)
    E = len(sc["est"])
    Re = torch.stack([e[0] for e in sc["est"]]).cuda()
    te = torch.stack([e[1] for e in sc["est"]]).cuda()
    info = torch.from_numpy(sc["info"])[None].repeat(E, 1, 1).cuda()
    err, ok = lib.registration_recall(Re, te, sc["rot"].repeat(E, 1, 1).cuda(), sc["trn"].repeat(E, 1, 1).cuda(), info, 0.2)
    # float64; the closed-form quaternion vs the eigenvector of nibabel's K matrix agree to rounding on rotation matrices
    assert np.allclose(err.cpu().numpy(), g[tag + "rr_err"], rtol=1e-6, atol=1e-12)
    assert ok.cpu().numpy().tolist() == [int(v) for v in g[tag + "rr_ok"]]


def test_tester_style_api_end_to_end():
    """the four calls of 3D/lib/tester.py:79-85 through the mirrored names"""
    from diffreg_hip.metrics import MatchMetrics as MML, evaluate_pairs
    sc = metrics_scene(256, 256, 3)
    data = dict(s_pcd=sc["s_pcd"].cuda(), t_pcd=sc["t_pcd"].cuda(), batched_rot=sc["rot"].cuda(), batched_trn=sc["trn"].cuda(),
                gt_cov=[sc["info"]], src_mask=torch.ones(1, 256, dtype=torch.bool).cuda(), tgt_mask=torch.ones(1, 256, dtype=torch.bool).cuda())
    mp = sc["matches"].cuda()
    rot, trn = MML.ransac_regist_coarse(data["s_pcd"], data["t_pcd"], data["src_mask"], data["tgt_mask"], mp, iters=5000)
    assert rot.dtype == torch.float64 and rot.shape == (1, 3, 3) and trn.shape == (1, 3, 1)
    ir = MML.compute_inlier_ratio(mp, data, inlier_thr=0.1).mean()
    rr = MML.compute_registration_recall(rot, trn, data, thr=0.2)
    want_ir = mo.inlier_ratio(sc["matches"], sc["s_pcd"], sc["t_pcd"], sc["rot"], sc["trn"], 0.1)
    assert float(ir) == float(want_ir[0]) and rr == 1.0
    from diffreg_hip.metrics import segment_matches
    seg, cnt = segment_matches(mp, 1)
    ev = evaluate_pairs(seg, cnt, data["s_pcd"], data["t_pcd"], data["batched_rot"], data["batched_trn"],
                        torch.from_numpy(sc["info"])[None].cuda(), ransac_iters=5000)
    assert float(ev["fmr"]) == 1.0 and int(ev["rr_ok"]) == 1 and torch.equal(ev["rot"], rot)


@pytest.mark.parametrize("name", ["patch64_k2_thr", "ragged_k3_or", "smallest_k1", "masked_k2"])
def test_mutual_topk_select_vs_reference_vectors(golden, name):
    """dr_mutual_topk_select_f32 against the reference's batch_mutual_topk_select (indices in torch.nonzero order, bit-exact)"""
    from diffreg_hip import lib
    from tests.helpers import topk_case
    g = golden("metrics_ref")
    c = topk_case(name)
    dv = lambda x: None if x is None else x.cuda()
    b, i, j, s = lib.batch_mutual_topk_select(c["score"].cuda(), c["k"], dv(c["row_masks"]), dv(c["col_masks"]), c["largest"], c["threshold"],
                                              c["mutual"])
    assert np.array_equal(torch.stack([b, i, j], 1).cpu().numpy(), g["mts_" + name + "_idx"])
    assert np.array_equal(s.cpu().numpy(), g["mts_" + name + "_score"])
    # the unbatched form (mutual_topk_select, vision3d/ops/mutual_topk_select.py:7-60) = one batch element
    i1, j1, s1 = lib.batch_mutual_topk_select(c["score"][0].cuda(), c["k"], None if c["row_masks"] is None else None, None, c["largest"],
                                              c["threshold"], c["mutual"])
    want = mo.batch_mutual_topk_select(c["score"][:1], c["k"], None, None, c["largest"], c["threshold"], c["mutual"])
    assert torch.equal(i1.cpu(), want[1]) and torch.equal(j1.cpu(), want[2]) and torch.equal(s1.cpu(), want[3])


@pytest.mark.parametrize("counts", [[0, 0, 0], [0, 5, 2], [1, 3, 64]])
def test_harness_with_no_or_few_matches(counts):
    """Pairs without matches (or fewer than a RANSAC sample) in a batch: zero inlier ratio / fitness and a finite pose, no launch failure
    (tester.py divides by max(len, 1); Open3D's RANSAC returns the identity when it cannot sample)."""
    from diffreg_hip import metrics
    torch.manual_seed(0)
    P, N, M, cap = 3, 40, 50, 64
    s, t = torch.randn(P, N, 3).cuda(), torch.randn(P, M, 3).cuda()
    R, tr = torch.eye(3)[None].repeat(P, 1, 1).cuda(), torch.zeros(P, 3, 1).cuda()
    matches = torch.zeros(P, cap, 3, dtype=torch.int64).cuda()
    for b, k in enumerate(counts):
        matches[b, :k, 0] = b
        matches[b, :k, 1] = torch.randint(0, N, (k,)).cuda()
        matches[b, :k, 2] = torch.randint(0, M, (k,)).cuda()
    out = metrics.evaluate_pairs(matches, torch.tensor(counts, dtype=torch.int32).cuda(), s, t, R, tr, ransac_iters=2000)
    torch.cuda.synchronize()
    for k in ("ir", "fitness", "rot", "trn"):
        assert torch.isfinite(out[k]).all(), k
    for b, k in enumerate(counts):
        if k == 0:
            assert float(out["ir"][b]) == 0.0 and float(out["fitness"][b]) == 0.0 and int(out["n_inlier"][b]) == 0
