"""oracle/train_oracle.py against the reference-minted vectors of the training forward (tests/golden/train_forward.npz, made by
oracle/make_golden_train.py from the reference's Pipeline.forward under model.train() and its MatchMotionLoss).  CPU only."""
import os

import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import train_oracle as tro
from tests.helpers import TRAIN_CASES, focal_case, train_case, train_weights

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_forward.npz"))


@pytest.mark.parametrize("tag", list(TRAIN_CASES))
def test_gt_noising_is_bit_exact(tag):
    c = train_case(tag)
    gt = tro.match_matrix(c["matches"], c["B"], c["N"], c["M"])
    x = tro.gt_noising(gt, c["randn"], c["ts"])
    assert x.dtype == torch.float64 and int(G[tag + "_ts"]) == c["ts"]
    assert np.array_equal(x.numpy(), G[tag + "_noised"])


@pytest.mark.parametrize("tag", list(TRAIN_CASES))
def test_losses_on_reference_matrices(tag):
    """focal loss, recall / precision and the L1 motion term from the reference's own conf matrices / match lists / (R, t)"""
    c = train_case(tag)
    gt = tro.match_matrix(c["matches"], c["B"], c["N"], c["M"])
    T = torch.from_numpy
    fc = tro.focal_loss(T(G[tag + "_conf_matrix_pred"]), gt)
    fh = tro.focal_loss(T(G[tag + "_conf_matrix_gt_hat"]), gt)
    r, p = tro.match_recall(gt, T(G[tag + "_coarse_match_pred"]))
    assert abs(float(fc) - float(G[tag + "_loss_mot0_focal_coarse"])) <= 1e-6 * abs(float(fc))
    assert abs(float(fh) - float(G[tag + "_loss_mot0_loss_matrix_gt_hat"])) <= 1e-6 * abs(float(fh))
    assert float(r) == float(G[tag + "_loss_mot0_recall_coarse"]) and float(p) == float(G[tag + "_loss_mot0_precision_coarse"])
    ov = torch.zeros(c["B"], c["N"], dtype=torch.bool)
    for b, m in enumerate(c["matches"]):
        ov[b][m[0]] = True
    l1 = tro.motion_l1(c["p_s"], T(G[tag + "_R_s2t_pred"]), T(G[tag + "_t_s2t_pred"]), c["R_gt"], c["t_gt"], ov)
    total = float(fc) + float(l1) + float(fh)
    assert abs(total - float(G[tag + "_loss_mot1_loss"])) <= 2e-6 * abs(total)
    assert abs(float(fc) + float(fh) - float(G[tag + "_loss_mot0_loss"])) <= 2e-6


@pytest.mark.parametrize("mt", ["sinkhorn", "dual_softmax"])
@pytest.mark.parametrize("gamma,alpha,pw,nw", [(2.0, 0.25, 1.0, 1.0), (1.5, 0.4, 0.7, 2.0)])
def test_focal_corner_cases(mt, gamma, alpha, pw, nw):
    conf, gt, weight, _ = focal_case()
    nm = "focal_%s_g%s" % (mt, str(gamma).replace(".", "p"))
    for suffix, g in (("", gt), ("_nopos", torch.zeros_like(gt)), ("_noneg", torch.ones_like(gt))):
        got = float(tro.focal_loss(conf, g, weight, alpha, gamma, pw, nw, mt))
        assert abs(got - float(G[nm + suffix])) <= 1e-6 * max(1e-3, abs(got)), (nm + suffix, got, float(G[nm + suffix]))


def test_match_recall_with_repeated_predictions():
    _, gt, _, _ = focal_case()
    r, p = tro.match_recall(gt, torch.from_numpy(G["recall_pred"]))
    assert float(r) == float(G["recall"]) and float(p) == float(G["precision"])


def test_training_forward_matches_reference():
    """the whole branch (coarse transformer with its positioning layer, coarse matching, Procrustes, noising, warp, denoiser,
    matching) restated from the pinned pieces of oracle/diffreg_oracle.py -- against the reference's outputs"""
    tag = "b1"
    c = train_case(tag)
    v = synth.VARIANTS["3dmatch"]
    o = tro.training_forward(train_weights(), v, c["f_s"], c["f_t"], c["p_s"], c["p_t"], c["src_mask"], c["tgt_mask"], c["matches"], c["randn"],
                             c["ts"], c["mc"])
    assert np.array_equal(o["noised"].numpy(), G[tag + "_noised"])
    assert np.abs(o["src_warped"].numpy() - G[tag + "_src_warped"]).max() < 1e-4
    assert np.abs(o["R_s2t_pred"].numpy() - G[tag + "_R_s2t_pred"]).max() < 1e-4
    for k in ("conf_matrix_pred", "conf_matrix_gt_hat"):
        d = np.abs(o[k].numpy() - G[tag + "_" + k])
        assert d.max() < 2e-3 and (d > 1e-4).mean() < 1e-3, (k, d.max())       # (host BLAS, sharp entries: as tests/test_oracle_golden.py)


GB = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_backward.npz"))


@pytest.mark.parametrize("tag", ["full", "masked", "big"])
def test_matching_head_backward_against_reference_autograd(tag):
    """the written-out adjoint recurrences against torch autograd through the reference's own log_optimal_transport and focal loss"""
    from tests.helpers import train_backward_case
    sc, gt, sm, tm = train_backward_case(tag)
    sc = sc.masked_fill(~(sm[:, :, None] & tm[:, None, :]), float("-inf"))
    conf = torch.from_numpy(GB[tag + "_conf"])
    gconf = tro.focal_loss_backward(conf, gt)
    assert np.abs(gconf.numpy() - GB[tag + "_grad_conf"]).max() <= 1e-6 * np.abs(GB[tag + "_grad_conf"]).max()
    gs, ga = tro.sinkhorn_backward(sc.double(), 1.0, 3, sm, tm, torch.from_numpy(GB[tag + "_grad_conf"]).double())
    ref = GB[tag + "_grad_scores"]
    assert np.abs(gs.numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    assert abs(float(ga) - float(GB[tag + "_grad_bin_score"])) <= 2e-5 * abs(float(GB[tag + "_grad_bin_score"])) + 1e-8
