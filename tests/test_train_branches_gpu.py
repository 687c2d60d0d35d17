"""The training graphs of the configuration branches no shipped yaml selects -- pe_type 'sinusoidal', entangled = True, match_type 'dual_softmax'
(3D/models/transformero.py:50-57, 234-254; matching.py:181, 193-205; loss.py:301-307) -- differentiable on the device
(diffreg_hip.autograd.coarse_branch / denoising_branch in the form the configuration selects), against torch autograd through the REFERENCE's own
modules in float32 and float64 (oracle/make_golden_train_branches.py -> tests/golden/train_backward_branches.npz): conf, loss, the gradients of the
backbone features and EVERY parameter gradient, entry-wise.  Soft head (logits O(10)): the bound is 1e-3 of each tensor's largest entry, or -- where
the reference's own float32 backward is further than that from float64 -- at least as close to float64 as the reference.

The case (tests/helpers.train_branch_case: one 40 x 32 pair) is evaluated at a feature scale the minting script chose so that no ReLU unit of either
graph is within 5e-6 of its kink (stored: <form>_feat_scale, <form>_relu_margin).  Gradients are only piecewise smooth: one hidden unit changing
sides moves every gradient upstream of it by a finite amount (measured on the first case tried: a unit 5.9e-8 from zero in layer 1 moved the
layer-0 / layer-1 gradients by 2e-3 of their maxima under a 1e-6 change of the sinusoidal code), so a point ON a kink compares nothing.  Needs a GPU."""
import os

import numpy as np
import pytest
import torch

from diffreg_hip import lib, synth
from tests.helpers import train_branch_case, train_weights
from tests.test_models_api_gpu import StubBackbone
from tests.test_branches_gpu import form_config
from tests.test_train_gpu import assert_gradient_entries

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_backward_branches.npz"))
STRIDE = int(G["stride"])
FORMS = {"sin": ("sinusoidal", False, "sinkhorn"), "rot_ent": ("rotary", True, "sinkhorn"), "sin_ent": ("sinusoidal", True, "sinkhorn"),
         "dsm": ("rotary", False, "dual_softmax")}


def psub(g):
    return (g[::STRIDE, ::STRIDE] if g.dim() == 2 else g).detach().cpu().numpy()


def build(form):
    from models.pipeline import Pipeline
    pe_type, ent, mtype = FORMS[form]
    c = train_branch_case(40, 32, 70)
    c["scale"] = float(G[form + "_feat_scale"])
    cfg = form_config("none", steps=20, mc=c["mc"], match_type=mtype)
    ct = cfg.coarse_transformer
    ct["pe_type"], ct["entangled"] = pe_type, ent
    ct["feature_matching"]["entangled"] = ent
    cfg.coarse_matching["entangled"] = ent
    if mtype == "dual_softmax":
        cfg.coarse_matching["dsmax_temperature"] = ct["feature_matching"]["dsmax_temperature"] = float(G["dsm_temperature"])
    model = Pipeline(cfg, backbone=StubBackbone())
    sd = model.state_dict()
    for k, a in train_weights("soft").items():
        if k in sd:
            sd[k] = a
    model.load_state_dict(sd)
    return c, model.to(DEV), mtype


def check_params(named, pre, n_expected):
    worst, checked = 0.0, 0
    for k, prm in named:
        key = pre + "g32_" + k
        if key in G.files:
            assert prm.grad is not None, k
            worst = max(worst, assert_gradient_entries(psub(prm.grad), G[key], G[key.replace("_g32_", "_g64_")], pre + " d/d " + k, 1e-3))
            checked += 1
        else:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
    assert checked == n_expected, (checked, n_expected)
    return worst


@pytest.mark.parametrize("form", list(FORMS))
def test_denoising_branch_backward_of_a_form(form):
    from diffreg_hip.autograd import denoising_branch, focal_loss
    c, model, mtype = build(form)
    fs = (c["f_s"] * c["scale"]).to(DEV).requires_grad_(True)
    ft = (c["f_t"] * c["scale"]).to(DEV).requires_grad_(True)
    warped = c["warped"].to(DEV)
    hat = denoising_branch(model, fs, ft, warped, c["p_t"].to(DEV), c["src_mask"].to(DEV), c["tgt_mask"].to(DEV))
    pre = form + "_branch_"
    assert np.abs(hat.detach().cpu().numpy() - G[pre + "conf32"]).max() <= 1e-4
    gt = torch.zeros_like(hat)
    gt[0][c["matches"][0][0].to(DEV), c["matches"][0][1].to(DEV)] = 1
    loss = focal_loss(hat, gt, match_type=mtype)
    loss.backward()
    assert abs(float(loss.detach()) - float(G[pre + "loss32"])) <= 1e-4 * float(G[pre + "loss32"])
    worst = 0.0
    for got, key in ((fs.grad, "grad_src"), (ft.grad, "grad_tgt")):
        worst = max(worst, assert_gradient_entries(got[0, ::3, ::4].cpu().numpy(), G[pre + key + "32"], G[pre + key + "64"], pre + key, 1e-3))
    named = list(model.denoising_transformer.named_parameters()) + [("head." + k, p) for k, p in model.denoising_coarse_matching.named_parameters()]
    worst = max(worst, check_params(named, pre, 61 if mtype == "dual_softmax" else 62))
    print("denoising branch, form %s: worst gradient deviation / tensor maximum %.2e" % (form, worst))


@pytest.mark.parametrize("form", list(FORMS))
def test_coarse_branch_backward_of_a_form(form):
    from diffreg_hip.autograd import coarse_branch, focal_loss, motion_l1
    c, model, mtype = build(form)
    fs = (c["f_s"] * c["scale"]).to(DEV).requires_grad_(True)
    ft = (c["f_t"] * c["scale"]).to(DEV).requires_grad_(True)
    ps, pt, sm, tm = c["p_s"].to(DEV), c["p_t"].to(DEV), c["src_mask"].to(DEV), c["tgt_mask"].to(DEV)
    conf, R, t = coarse_branch(model, fs, ft, ps, pt, sm, tm)
    pre = form + "_coarse_"
    assert np.abs(conf.detach().cpu().numpy() - G[pre + "conf32"]).max() <= 1e-4
    assert np.abs(R.detach().cpu().numpy() - G[pre + "R32"]).max() < 1e-4 and np.abs(t.detach().cpu().numpy() - G[pre + "t32"]).max() < 1e-4
    gt = torch.zeros_like(conf)
    gt[0][c["matches"][0][0].to(DEV), c["matches"][0][1].to(DEV)] = 1
    ov = torch.zeros(1, c["N"], dtype=torch.bool, device=DEV)
    ov[0][c["matches"][0][0].to(DEV)] = True
    loss = focal_loss(conf, gt, match_type=mtype) + 0.1 * motion_l1(ps, R, t, c["R_gt"].to(DEV), c["t_gt"].to(DEV), ov)
    loss.backward()
    assert abs(float(loss.detach()) - float(G[pre + "loss32"])) <= 1e-3 * float(G[pre + "loss32"])
    worst = 0.0
    for got, key in ((fs.grad, "grad_src"), (ft.grad, "grad_tgt")):
        worst = max(worst, assert_gradient_entries(got[0, ::3, ::4].cpu().numpy(), G[pre + key + "32"], G[pre + key + "64"], pre + key, 1e-3))
    named = list(model.coarse_transformer.named_parameters()) + [("head." + k, p) for k, p in model.coarse_matching.named_parameters()]
    worst = max(worst, check_params(named, pre, 41 if mtype == "dual_softmax" else 42))
    print("coarse branch, form %s: worst gradient deviation / tensor maximum %.2e" % (form, worst))


@pytest.mark.parametrize("masked", [False, True])
def test_dual_softmax_backward_against_torch(masked):
    """dr_dual_softmax_backward_f32 against torch autograd through the reference's arithmetic (matching.py:193-205) in float64"""
    torch.manual_seed(3)
    P, N, M, T = 2, 70, 90, 0.7
    sim = torch.randn(P, N, M, dtype=torch.float64) * 2
    g = torch.randn(P, N, M, dtype=torch.float64)
    sm = (torch.arange(N)[None] < torch.tensor([[60], [70]])) if masked else None
    tm = (torch.arange(M)[None] < torch.tensor([[90], [75]])) if masked else None
    s = sim.clone().requires_grad_(True)
    s1 = s / T
    if masked:
        s2 = s1.clone()
        s1 = s1.masked_fill(~sm[:, :, None], float("-inf"))
        s2 = s2.masked_fill(~tm[:, None, :], float("-inf"))
        conf = torch.softmax(s1, 1) * torch.softmax(s2, 2)
    else:
        conf = torch.softmax(s1, 1) * torch.softmax(s1, 2)
    (conf * g).sum().backward()
    dev = lambda t_: None if t_ is None else t_.to(DEV)
    got_c = lib.dual_softmax(sim.float().to(DEV), T, dev(sm), dev(tm))
    assert float((got_c.double().cpu() - conf.detach()).abs().max()) < 1e-6
    got = lib.dual_softmax_backward(sim.float().to(DEV), T, dev(sm), dev(tm), g.float().to(DEV))
    assert float((got.double().cpu() - s.grad).abs().max()) < 2e-6 * float(s.grad.abs().max()) + 1e-9


@pytest.mark.parametrize("form", ["sin", "rot_ent", "sin_ent"])
def test_whole_training_step_of_a_form(form):
    """Pipeline.forward_train + MatchMotionLoss.forward_train with a non-default form selected (stub backbone): the loss equals the value-only path's
    (Pipeline.forward under .train() + MatchMotionLoss.forward: the overlay's modules, pinned to the reference by tests/test_branches_gpu.py), and
    .backward() reaches every parameter the reference trains behind the backbone -- 104 tensors (the entangled forms skip the positioning layer,
    transformero.py:252, whose Matching has no gradient in any form)."""
    from models.loss import MatchMotionLoss
    from models.pipeline import Pipeline
    from tests.test_train_gpu import LOSS_CFG
    from tests.helpers import train_case
    pe_type, ent, mtype = FORMS[form]
    c = train_case("b1")
    B, N, M = c["B"], c["N"], c["M"]
    cfg = form_config("none", steps=20, mc=c["mc"])
    ct = cfg.coarse_transformer
    ct["pe_type"], ct["entangled"] = pe_type, ent
    ct["feature_matching"]["entangled"] = ent
    cfg.coarse_matching["entangled"] = ent
    model = Pipeline(cfg, backbone=StubBackbone())
    sd = model.state_dict()
    for k, a in train_weights("soft").items():
        sd[k] = a
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    feats = torch.cat([c["f_s"].reshape(B * N, -1), c["f_t"].reshape(B * M, -1)], 0) * 0.5
    pts = torch.cat([c["p_s"].reshape(B * N, 3), c["p_t"].reshape(B * M, 3)], 0)

    def batch():
        return {"points": [None, None, pts.to(DEV), None], "_feats": feats.to(DEV), "src_mask": c["src_mask"].to(DEV), "tgt_mask": c["tgt_mask"].to(DEV),
                "src_ind_coarse_split": torch.arange(B * N, device=DEV), "tgt_ind_coarse_split": torch.arange(B * M, device=DEV),
                "src_ind_coarse": torch.arange(B * N, device=DEV), "tgt_ind_coarse": torch.arange(B * N, B * (N + M), device=DEV),
                "coarse_matches": [m.to(DEV) for m in c["matches"]], "batched_rot": c["R_gt"].to(DEV), "batched_trn": c["t_gt"].to(DEV),
                "ts": torch.tensor([c["ts"]], device=DEV), "randn": c["randn"].to(DEV)}
    crit = MatchMotionLoss(dict(LOSS_CFG, motion_weight=0.1))
    with torch.no_grad():
        ref_info = crit(model(batch()))
    info = crit.forward_train(model.forward_train(batch()))
    assert abs(float(info["loss"].detach()) - float(ref_info["loss"])) <= 1e-4 * float(ref_info["loss"])
    info["loss"].backward()
    with_grad = [k for k, p in model.named_parameters() if p.grad is not None and bool(torch.isfinite(p.grad).all())
                 and (float(p.grad.abs().max()) > 0 or k.endswith("bin_score"))]
    assert len(with_grad) == 104, (len(with_grad), sorted(set(k for k, _ in model.named_parameters()) - set(with_grad))[:8])
    assert not any(k.startswith("coarse_transformer.layers.2.") or k.endswith("tgt_proj.weight") for k in with_grad)
