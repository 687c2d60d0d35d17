"""CPU: the evaluation-harness oracle (oracle/metrics_oracle.py) against vectors minted by the reference's own functions
(oracle/make_golden_metrics.py -> tests/golden/metrics_ref.npz), and the properties that anchor the RANSAC restatement."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import metrics_oracle as mo
from tests.helpers import metrics_scene

CASES = [(256, 256, 3), (200, 256, 5), (512, 512, 8)]


@pytest.mark.parametrize("N,M,seed", CASES)
def test_oracle_matches_reference_vectors(golden, N, M, seed):
    g = golden("metrics_ref")
    tag = "%dx%d_s%d_" % (N, M, seed)
    sc = metrics_scene(N, M, seed)
    ir3 = mo.inlier_ratio(sc["matches"], sc["s_pcd"], sc["t_pcd"], sc["rot"], sc["trn"], 0.1)
    assert np.array_equal(ir3.numpy(), g[tag + "ir3d"])
    ir4 = mo.inlier_ratio(sc["matches"], sc["s_pcd"], sc["t_pcd4"], sc["rot"], sc["trn"], 0.04, sc["coarse_flow"][None])
    assert np.array_equal(ir4.numpy(), g[tag + "ir4d"])
    val, per, blends = mo.nrfmr(sc["matches"], sc["s_pcd"], sc["t_pcd4"], [sc["raw_pcd"]], [sc["raw_flow"]], [sc["metric_index"]],
                                sc["rot"], sc["trn"], 0.04)
    assert np.array_equal(blends[0], g[tag + "blended"])
    assert float(val) == float(g[tag + "nrfmr"])
    for k, (Re, te) in enumerate(sc["est"]):
        rr, errs = mo.registration_recall(Re[None], te[None], sc["rot"], sc["trn"], [sc["info"]], 0.2)
        assert errs[0] == pytest.approx(g[tag + "rr_err"][k], rel=1e-12, abs=1e-15)
        assert rr == g[tag + "rr_ok"][k]


def test_stable_tie_rule_differs_only_on_tied_queries():
    """matches that share a source point are anchors at the same place: the reference's 3-NN choice among them is whatever
    np.argpartition leaves; the kernel's rule (lowest match row first) gives the same blend wherever no tie reaches the
    third neighbour"""
    sc = metrics_scene(256, 256, 3)
    m = sc["matches"]
    anchors = sc["s_pcd"][0][m[:, 1]].numpy()
    motion = (sc["t_pcd4"][0][m[:, 2]] - sc["s_pcd"][0][m[:, 1]]).numpy()
    q = sc["raw_pcd"][sc["metric_index"]].numpy()
    a, _ = mo.blend_anchor_motion(q, anchors, motion)
    b, _ = mo.blend_anchor_motion(q, anchors, motion, stable=True)
    d = np.sort(np.sum((anchors[None] - q[:, None]) ** 2, -1), 1)
    untied = (d[:, 0] != d[:, 1]) & (d[:, 1] != d[:, 2]) & (d[:, 2] != d[:, 3])
    assert untied.sum() > 100 and np.array_equal(a[untied], b[untied])


def test_mat2quat_identities():
    # the nibabel step has no reference vector (nibabel absent): check it as a quaternion of the rotation
    for k in range(20):
        R = synth._rodrigues(synth.hash_normal(k, 1, (3,)), float(synth.hash_uniform(k, 2, (1,), -3.1, 3.1)[0]))
        w, x, y, z = mo.mat2quat(R)
        assert w >= 0 and abs(w * w + x * x + y * y + z * z - 1) < 1e-12
        back = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.abs(back - R).max() < 1e-12


def test_hash_bits_is_the_integer_under_hash_u01():
    b = synth.hash_bits(5, 9, 1000)
    assert np.array_equal((b.astype(np.float64) + 0.5) / float(1 << 53), synth.hash_u01(5, 9, 1000))


def _corr(sc):
    return sc["matches"][:, 1:].numpy()


def test_ransac_recovers_the_generating_pose():
    sc = metrics_scene(256, 256, 3)
    r = mo.ransac_corr(sc["s_pcd"][0].numpy(), sc["t_pcd"][0].numpy(), _corr(sc), 0.05, iters=4000, seed=1)
    assert np.abs(r["R"] - sc["pair"]["R_gt"]).max() < 0.02 and np.abs(r["t"] - sc["pair"]["t_gt"]).max() < 0.05
    assert abs(np.linalg.det(r["R"]) - 1) < 1e-12 and np.abs(r["R"] @ r["R"].T - np.eye(3)).max() < 1e-12
    # fitness = a recount of the inliers under the returned pose
    S = sc["s_pcd"][0].numpy().astype(np.float64)[_corr(sc)[:, 0]]
    Y = sc["t_pcd"][0].numpy().astype(np.float64)[_corr(sc)[:, 1]]
    d2 = ((S @ r["R"].T + r["t"] - Y) ** 2).sum(1)
    assert (d2 < 0.05 ** 2).sum() == r["n_inlier"] == round(r["fitness"] * len(S))
    assert r["inlier_rmse"] == pytest.approx(np.sqrt(d2[d2 < 0.05 ** 2].mean()), rel=1e-9)
    # more iterations never lower the score: the first 1000 hypotheses are a prefix of the 4000
    r1 = mo.ransac_corr(sc["s_pcd"][0].numpy(), sc["t_pcd"][0].numpy(), _corr(sc), 0.05, iters=1000, seed=1)
    assert r1["n_inlier"] <= r["n_inlier"]


def test_ransac_degenerate_inputs():
    sc = metrics_scene(200, 256, 5)
    s, t = sc["s_pcd"][0].numpy(), sc["t_pcd"][0].numpy()
    r = mo.ransac_corr(s, t, _corr(sc)[:2], 0.05, iters=100)
    assert np.array_equal(r["R"], np.eye(3)) and r["best_iter"] == -1          # < 3 matches: identity (loss.py:363-366)
    same_src = np.stack([np.zeros(10, np.int64), np.arange(10)], 1)
    r = mo.ransac_corr(s, t, same_src, 0.05, iters=100)
    assert r["best_iter"] == -1                                                  # every triple repeats a source point


def test_segment_matches_layout():
    from diffreg_hip.metrics import segment_matches
    m = torch.tensor([[0, 1, 2], [0, 3, 4], [2, 5, 6], [2, 7, 8], [2, 9, 1]])
    seg, cnt = segment_matches(m, 3)
    assert cnt.tolist() == [2, 0, 3] and seg.shape == (3, 5, 3)
    assert torch.equal(seg[0, :2], m[:2]) and torch.equal(seg[2, :3], m[2:])


@pytest.mark.parametrize("name", ["patch64_k2_thr", "ragged_k3_or", "smallest_k1", "masked_k2"])
def test_mutual_topk_oracle_vs_reference_vectors(golden, name):
    from tests.helpers import topk_case
    g = golden("metrics_ref")
    c = topk_case(name)
    b, i, j, s = mo.batch_mutual_topk_select(c["score"], c["k"], c["row_masks"], c["col_masks"], c["largest"], c["threshold"], c["mutual"])
    assert np.array_equal(torch.stack([b, i, j], 1).numpy(), g["mts_" + name + "_idx"])
    assert np.array_equal(s.numpy(), g["mts_" + name + "_score"]) and len(b) > 0
