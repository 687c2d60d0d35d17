"""the KPFCN backbone ops and the whole coarse phase through the C ABI, against the oracle and against
the output of the reference backbone itself (tests/golden/kpfcn_coarse.npz).  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import kpfcn_oracle as ko
from tests.helpers import T
from tests.test_oracle_golden import kpfcn_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_kpconv_matches_oracle(golden):
    from diffreg_hip import lib
    g, sd, tb = kpfcn_inputs(golden)
    cfg = synth.KPFCN_CFG
    # layer-1 geometry with random features of 64 channels, then a strided one (pools), then Cin = 1
    for q, s, idx, cin, layer in ((tb["points"][1], tb["points"][1], tb["neighbors"][1], 64, 1),
                                  (tb["points"][2], tb["points"][1], tb["pools"][1], 96, 1),
                                  (tb["points"][0], tb["points"][0], tb["neighbors"][0], 1, 0)):
        x = T(synth.hash_normal(9, cin + len(s), (len(s), cin))).float()
        Wk = T(synth.hash_uniform(10, cin, (cfg["num_kernel_points"], cin, 48))).float() / (cin * 4) ** 0.5
        kp = sd["encoder_blocks.3.KPConv.kernel_points"] if layer == 1 else sd["encoder_blocks.0.KPConv.kernel_points"]
        extent = cfg["first_subsampling_dl"] * 2 ** layer * cfg["KP_extent"]
        ref = ko.kpconv(q, s, idx, x, Wk, kp, extent)
        wf = lib.kpconv_gather(q.to(DEV), s.to(DEV), idx.to(DEV), x.to(DEV), kp.to(DEV), extent)
        w2 = Wk.permute(2, 0, 1).reshape(48, -1)
        w2 = torch.cat([w2, torch.zeros(48, (-w2.shape[1]) % 4)], 1).contiguous()
        got = lib.linear_ex(wf, w2.to(DEV)).cpu()
        assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_norm_and_pools_match_oracle():
    from diffreg_hip import lib
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(700, 96, generator=g) * 3 + 1, torch.randn(700, 96, generator=g)
    sa, sb = lib.col_stats(a.to(DEV)), lib.col_stats(b.to(DEV))
    import torch.nn.functional as F
    assert (lib.norm_apply(a.to(DEV), sa).cpu() - F.leaky_relu(ko.norm_block(a), 0.1)).abs().max().item() < 2e-5
    assert (lib.norm_apply(a.to(DEV), sa, activate=False).cpu() - ko.norm_block(a)).abs().max().item() < 2e-5
    ref = F.leaky_relu(ko.norm_block(a) + ko.norm_block(b), 0.1)
    assert (lib.norm_apply(a.to(DEV), sa, b.to(DEV), sb).cpu() - ref).abs().max().item() < 2e-5
    ref = F.leaky_relu(ko.norm_block(a) + b, 0.1)
    assert (lib.norm_apply(a.to(DEV), sa, b.to(DEV), None).cpu() - ref).abs().max().item() < 2e-5
    inds = torch.randint(0, 701, (300, 17), generator=g)            # 700 = shadow index
    assert torch.equal(lib.gather_pool(a.to(DEV), inds.to(DEV)).cpu(), ko.max_pool(a, inds))
    assert torch.equal(lib.gather_pool(a.to(DEV), inds.to(DEV), first_only=True).cpu(), ko.closest_pool(a, inds))


def test_kpfcn_coarse_matches_reference(golden):
    """the whole coarse phase on the GPU against the reference backbone's own output"""
    from diffreg_hip.backbone import KPFCNEngine
    g, sd, tb = kpfcn_inputs(golden)
    eng = KPFCNEngine(sd, device=DEV)
    out = eng.forward(tb).cpu().numpy()
    assert out.shape == g["coarse"].shape
    err = np.abs(out - g["coarse"]).max()
    assert err < 1e-4 * max(1.0, np.abs(g["coarse"]).max()), err       # north_star tolerance: 1e-4 fp32


def test_models_backbone_overlay_runs_the_engine(golden):
    """models.backbone.KPFCN (the overlay module with the reference's state-dict layout) loads the weights and returns
    the reference's coarse features through forward(batch, phase='coarse')."""
    import importlib.util, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("dr_models_backbone", os.path.join(here, "..", "diff-reg_amd", "models", "backbone.py"))
    mb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mb)
    g, sd, tb = kpfcn_inputs(golden)
    cfg = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum", deformable=False,
               use_batch_norm=True, fine_feature_dim=264)
    net = mb.KPFCN(cfg).eval()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(not m.startswith(("encoder_blocks.", "decoder_blocks.1.", "coarse_out.")) for m in missing)
    dev_batch = {k: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV) for k, v in tb.items()}
    out = net.to(DEV)(dev_batch, phase="coarse").cpu().numpy()
    assert np.abs(out - g["coarse"]).max() < 1e-4 * max(1.0, np.abs(g["coarse"]).max())
    with pytest.raises(NotImplementedError):
        net(dev_batch, phase="fine")


def test_backward_ops_match_autograd(golden):
    """row f3, second half: the backward kernels of the backbone's ops (csrc/backbone_bwd.hip) against torch autograd through the oracle's
    restatement of the same ops: KPConv's gather half (scatter of the influences, normalised by the forward's neighbour count), the fused
    normalisation + LeakyReLU + residual sum in its three forms, max_pool / closest_pool."""
    from diffreg_hip import lib
    import torch.nn.functional as F
    g, sd, tb = kpfcn_inputs(golden)
    cfg = synth.KPFCN_CFG
    for q, s, idx, cin, layer in ((tb["points"][1], tb["points"][1], tb["neighbors"][1], 64, 1), (tb["points"][2], tb["points"][1], tb["pools"][1], 96, 1),
                                  (tb["points"][0], tb["points"][0], tb["neighbors"][0], 1, 0), (tb["points"][2], tb["points"][2], tb["neighbors"][2], 200, 2)):
        x = T(synth.hash_normal(9, cin + len(s), (len(s), cin))).float().requires_grad_(True)
        Wk = (T(synth.hash_uniform(10, cin, (cfg["num_kernel_points"], cin, 48))).float() / (cin * 4) ** 0.5).requires_grad_(True)
        kp = sd["encoder_blocks.3.KPConv.kernel_points"] if layer >= 1 else sd["encoder_blocks.0.KPConv.kernel_points"]
        extent = cfg["first_subsampling_dl"] * 2 ** layer * cfg["KP_extent"]
        ref = ko.kpconv(q, s, idx, x, Wk, kp, extent)
        G = T(synth.hash_normal(11, cin, tuple(ref.shape))).float()
        (ref * G).sum().backward()
        # the library: gather (+ 1 / num) then ONE GEMM; backward = the GEMM's two products + the gather's scatter
        w2 = Wk.detach().permute(2, 0, 1).reshape(48, -1)
        w2 = torch.cat([w2, torch.zeros(48, (-w2.shape[1]) % 4)], 1).contiguous().to(DEV)
        g_wf = lib.linear(G.to(DEV), w2.t().contiguous())                      # G W2  [Nq, ceil4(K Cin)]
        gx = lib.kpconv_gather_backward(q.to(DEV), s.to(DEV), idx.to(DEV), x.detach().to(DEV), kp.to(DEV), extent, g_wf).cpu()
        assert (gx - x.grad).abs().max().item() <= 2e-5 * max(1.0, x.grad.abs().max().item()), (cin, layer)
    gen = torch.Generator().manual_seed(3)
    a = (torch.randn(700, 96, generator=gen) * 3 + 1).requires_grad_(True)
    b = torch.randn(700, 96, generator=gen).requires_grad_(True)
    G = torch.randn(700, 96, generator=gen)
    for mode in ("a", "a_noact", "ab_norm", "ab_id"):
        a.grad = b.grad = None
        if mode == "a":
            ref = F.leaky_relu(ko.norm_block(a), 0.1)
        elif mode == "a_noact":
            ref = ko.norm_block(a)
        elif mode == "ab_norm":
            ref = F.leaky_relu(ko.norm_block(a) + ko.norm_block(b), 0.1)
        else:
            ref = F.leaky_relu(ko.norm_block(a) + b, 0.1)
        (ref * G).sum().backward()
        ad, bd = a.detach().to(DEV), b.detach().to(DEV)
        sa, sb = lib.col_stats(ad), lib.col_stats(bd)
        has_b, nb, act = mode.startswith("ab"), mode == "ab_norm", mode != "a_noact"
        out = lib.norm_apply(ad, sa, bd if has_b else None, sb if nb else None, activate=act)
        ga, gb = lib.norm_backward(G.to(DEV), out, ad, sa, bd if has_b else None, sb if nb else None, activate=act)
        assert (ga.cpu() - a.grad).abs().max().item() <= 2e-5 * a.grad.abs().max().item(), mode
        if has_b:
            assert (gb.cpu() - b.grad).abs().max().item() <= 2e-5 * b.grad.abs().max().item(), mode
    inds = torch.randint(0, 701, (300, 17), generator=gen)
    for first in (False, True):
        a.grad = None
        ref = ko.closest_pool(a, inds) if first else ko.max_pool(a, inds)
        Gp = torch.randn(ref.shape, generator=gen)
        (ref * Gp).sum().backward()
        got = lib.gather_pool_backward(a.detach().to(DEV), inds.to(DEV), Gp.to(DEV), first_only=first).cpu()
        assert (got - a.grad).abs().max().item() <= 1e-5 * max(1.0, a.grad.abs().max().item()), first


def test_kpfcn_backward_matches_reference(golden):
    """The whole coarse phase under autograd on the device (models.backbone.KPFCN in .train(): diffreg_hip/backbone_autograd.py) against the
    gradients autograd through the REFERENCE backbone produced for the same loss (tests/golden/kpfcn_coarse.npz, oracle/make_golden_kpfcn.py):
    every one of the 38 parameter tensors, norm and 256 sampled entries, to 1e-4."""
    import importlib.util, os
    from tests.test_oracle_golden import assert_gradients_match_reference
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("dr_models_backbone", os.path.join(here, "..", "diff-reg_amd", "models", "backbone.py"))
    mb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mb)
    g, sd, tb = kpfcn_inputs(golden)
    cfg = dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum", deformable=False,
               use_batch_norm=True, fine_feature_dim=264)
    net = mb.KPFCN(cfg)
    net.load_state_dict(sd, strict=False)
    net = net.to(DEV).train()
    dev_batch = {k: [t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV) for k, v in tb.items()}
    out = net(dev_batch, phase="coarse")
    assert out.requires_grad and np.abs(out.detach().cpu().numpy() - g["coarse"]).max() < 1e-4 * max(1.0, np.abs(g["coarse"]).max())
    G = T(synth.hash_normal(77, 1, tuple(out.shape)).astype(np.float32)).to(DEV)
    (out * G).sum().backward()
    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    assert_gradients_match_reference(grads, g)
    # and an optimiser step moves every one of them
    before = {k: p.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    torch.optim.SGD([p for p in net.parameters() if p.grad is not None], lr=1e-3).step()
    assert all(not torch.equal(before[k], dict(net.named_parameters())[k].detach()) for k in before)
