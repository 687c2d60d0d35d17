"""The KPConv / BatchNormBlock options no shipped yaml selects -- KP_influence 'constant' / 'gaussian', aggregation_mode 'closest', use_batch_norm = False
(3D/models/blocks.py:304-326, 430-446) -- against vectors minted by the reference backbone with each option set (oracle/make_golden_kpfcn_variants.py):
the restatement on the CPU, and on the GPU the engine's forward, the overlay module's forward and the whole coarse phase under autograd."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from diffreg_hip import synth
from tests.helpers import T

VARIANTS = {"gauss_sum_bn": ("gaussian", "sum", True), "const_closest_bn": ("constant", "closest", True), "linear_sum_nobn": ("linear", "sum", False),
            "gauss_closest_nobn": ("gaussian", "closest", False)}
USED = ("encoder_blocks.", "decoder_blocks.1.", "coarse_out.")
DEV = "cuda:0"


def overlay_module():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("dr_models_backbone", os.path.join(here, "..", "diff-reg_amd", "models", "backbone.py"))
    mb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mb)
    return mb


def variant_inputs(golden, tag):
    g = golden("kpfcn_variants")
    influence, aggregation, use_bn = VARIANTS[tag]
    kp = {k[len(tag) + 4:]: g[k] for k in g.files if k.startswith(tag + ":kp:")}
    sd = synth.make_kpfcn_weights(kp)
    cfg = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence=influence, aggregation_mode=aggregation, deformable=False,
               use_batch_norm=use_bn, fine_feature_dim=264)
    net = overlay_module().KPFCN(cfg)
    if not use_bn:               # the biases of the BatchNormBlocks: the overlay module has the reference's parameter names
        sd.update(synth.make_kpfcn_bn_biases([(k, tuple(v.shape)) for k, v in net.state_dict().items()
                                              if k.startswith(USED) and ".batch_norm" in k and k.endswith(".bias")]))
    sd = {k: T(v) for k, v in sd.items()}
    b = synth.make_kpfcn_batch()
    tb = dict(points=[T(p) for p in b["points"]], neighbors=[T(p) for p in b["neighbors"]], pools=[T(p) for p in b["pools"]],
              upsamples=[T(p) for p in b["upsamples"]], features=T(b["features"]))
    return g, cfg, net, sd, tb


def check_gradients(g, tag, grads, tol=1e-3):
    """256 sampled entries of every parameter gradient: within `tol` = 1e-3 of the tensor's largest entry of the float64 evaluation (row f3's bar for
    gradients, tests/test_train_gpu.py), or -- where the reference's own float32 backward is further than that from float64: the InstanceNorm forms,
    up to 2.5e-3 -- as close to float64 as the reference is (x 1.5).  Measured: InstanceNorm forms 2.8e-6 of the tensor maximum at worst where the
    reference's float32 is 2.5e-3 (its weight gradient is ONE float32 sum over ~8 000 points; the device's used to equal it to four digits -- same
    order, same rounding -- until the contraction became two-level: chunks of 1 024 points on the MFMA, partials summed in float64); without
    InstanceNorm 5.4e-3 = the reference's 5.4e-3 (linear / sum) and 6.2e-4 against 1.8e-4 (gaussian / closest), both on a BatchNormBlock bias: the
    float32 FORWARD of unnormalised activations, not a sum of the backward."""
    keys = [str(k) for k in g[tag + ":grad_keys"]]
    assert len(keys) >= 38
    worst = (0.0, 0.0, "")
    for k in keys:
        idx, val, gmax, val64 = g["%s:gidx:%s" % (tag, k)], g["%s:gval:%s" % (tag, k)], float(g["%s:gmax:%s" % (tag, k)]), g["%s:g64val:%s" % (tag, k)]
        smp = grads[k].detach().double().reshape(-1).cpu()[torch.from_numpy(idx)].numpy()
        e_ref = float(np.abs(val - val64).max())
        assert np.abs(smp - val64).max() <= max(tol * gmax, 1.5 * e_ref), (k, float(np.abs(smp - val64).max() / gmax), e_ref / gmax)
        assert np.abs(smp - val).max() <= tol * gmax + 2.5 * e_ref, (k, float(np.abs(smp - val).max() / gmax))
        worst = max(worst, (float(np.abs(smp - val64).max() / gmax), e_ref / gmax, k))
    print("%s: largest gradient deviation from float64 / tensor maximum %.2e (the reference's own float32 there: %.2e) on %s" % ((tag,) + worst))


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_variant_oracle_matches_reference(golden, tag):
    from oracle import kpfcn_oracle as ko
    g, cfg, net, sd, tb = variant_inputs(golden, tag)
    ocfg = dict(synth.KPFCN_CFG, KP_influence=cfg["KP_influence"], aggregation_mode=cfg["aggregation_mode"], use_batch_norm=cfg["use_batch_norm"])
    out = ko.kpfcn_coarse(sd, tb, cfg=ocfg)
    ref = g[tag + ":coarse"]
    assert np.abs(out.numpy()[::2] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(VARIANTS))
def test_variant_forward_and_backward_on_the_device(golden, tag):
    g, cfg, net, sd, tb = variant_inputs(golden, tag)
    ref = g[tag + ":coarse"]
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(not m.startswith(USED) for m in missing), (unexpected, [m for m in missing if m.startswith(USED)])
    dev_batch = {k: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV) for k, v in tb.items()}
    net = net.to(DEV).eval()
    out = net(dev_batch, phase="coarse").cpu().numpy()
    assert np.abs(out[::2] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    net.train()
    out = net(dev_batch, phase="coarse")
    assert out.requires_grad and np.abs(out.detach().cpu().numpy()[::2] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    G = T(synth.hash_normal(77, 1, tuple(out.shape)).astype(np.float32)).to(DEV)
    (out * G).sum().backward()
    check_gradients(g, tag, {k: p.grad for k, p in net.named_parameters() if p.grad is not None})


def test_deformable_kernels_are_refused():
    cfg = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum", deformable=True, use_batch_norm=True,
               fine_feature_dim=264)
    with pytest.raises(NotImplementedError):
        overlay_module().KPFCN(cfg)
    with pytest.raises(ValueError):
        overlay_module().KPFCN(dict(cfg, deformable=False, KP_influence="cubic"))


@pytest.mark.gpu
def test_weight_gradient_contraction_is_two_level():
    """g^T x over thousands of points (the KPConv / unary weight gradients): chunks of 1 024 points on the MFMA, partial products summed in float64
    -- against the float64 product, and against the one-chain float32 contraction it replaced (which is what a sequential float32 sum gives)"""
    from diffreg_hip import backbone_autograd as ba
    gen = torch.Generator().manual_seed(3)
    R = 9000
    g = (torch.randn(R, 64, generator=gen) + 0.5).to(DEV)
    x = (torch.randn(R, 40, generator=gen) + 2.0).to(DEV)            # (a mean: the running sum grows, its rounding with it)
    ref = g.double().t() @ x.double()
    two = ba._weight_grad(g, x).double()
    one = ba._mm(ba._tr(g), ba._tr(x)).double()
    M_ = float(ref.abs().max())
    e2, e1 = float((two - ref).abs().max()) / M_, float((one - ref).abs().max()) / M_
    assert e2 <= 2e-7 and e2 < e1, (e2, e1)
    small = ba._weight_grad(g[:1500], x[:1500]).double()            # at most two chunks: the single launch
    assert float((small - g[:1500].double().t() @ x[:1500].double()).abs().max()) <= 1e-5 * M_
