"""The configuration branches no shipped yaml selects -- pe_type 'sinusoidal', entangled = True, match_type 'dual_softmax'
(3D/models/position_encoding.py:43-44, 68-69; transformero.py:50-57, 234-254; matching.py:181, 193-205) -- oracle against vectors minted by the
reference (oracle/make_golden_branches.py).  CPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, masks

FORMS = {"sin": ("sinusoidal", False), "rot_ent": ("rotary", True), "sin_ent": ("sinusoidal", True)}


def soft_weights():
    from tests.helpers import train_weights
    return train_weights("soft")           # denoising_* and coarse_* tensors of the soft head (HEAD_GAIN_SOFT), as the mint script builds them


def scene():
    v = synth.VARIANTS["3dmatch"]
    p = synth.make_pair(64, 48, v["C"], seed=3)
    return dict(f_s=T(p["src_feats"])[None] * 0.5, f_t=T(p["tgt_feats"])[None] * 0.5, p_s=T(p["s_pcd"])[None], p_t=T(p["t_pcd"])[None],
                R_gt=T(p["R_gt"]).float(), t_gt=T(p["t_gt"]).float())


def form_cfg(tag, **kw):
    pe_type, ent = FORMS[tag]
    return dict(synth.VARIANTS["3dmatch"], pe_type=pe_type, entangled=ent, **kw)


def test_sinusoidal_position_code(golden):
    g = golden("3dmatch_branches")
    v = synth.VARIANTS["3dmatch"]
    code = orc.vol_pe_sinusoidal(scene()["p_s"], v["C"], v["origin"], v["voxel"])
    assert code.shape == (1, 64, v["C"])
    assert np.abs(code[0, :16].numpy() - g["pe_sin"]).max() < 2e-6


@pytest.mark.parametrize("tag", list(FORMS))
def test_layer_denoiser_and_head_of_every_form(golden, tag):
    g = golden("3dmatch_branches")
    W, sc, cfg = soft_weights(), scene(), form_cfg(tag)
    pre = "denoising_transformer.layers.1."
    sm_p, tm_p = masks(64, 48, 50, 41)
    sm_f, tm_f = masks(64, 48)
    pes, pet = (None, None) if cfg["entangled"] else (orc.position_code(cfg, sc["p_s"]), orc.position_code(cfg, sc["p_t"]))
    out = orc.attention_layer(W, pre, sc["f_s"], sc["f_t"], pes, pet, sm_p, tm_p, cfg["H"])
    assert np.abs(out[0].numpy() - g[tag + "_layer_cross_mask"]).max() < 2e-5
    out = orc.attention_layer(W, pre, sc["f_s"], sc["f_s"], pes, pes, sm_f, sm_f, cfg["H"])
    assert np.abs(out[0].numpy() - g[tag + "_layer_self_full"]).max() < 2e-5
    for mtag, (sm, tm) in (("full", (sm_f, tm_f)), ("mask", (sm_p, tm_p))):
        hs, ht, pe_s, pe_t = orc.denoiser(W, cfg, sc["f_s"], sc["f_t"], sc["p_s"], sc["p_t"], sm, tm)
        assert np.abs(hs[0].numpy() - g["%s_den_fs_%s" % (tag, mtag)]).max() < 1e-4
        assert np.abs(ht[0].numpy() - g["%s_den_ft_%s" % (tag, mtag)]).max() < 1e-4
        conf = orc.match_head(W, cfg, hs, ht, pe_s, pe_t, sm, tm)
        ref = g["%s_conf_%s" % (tag, mtag)]
        assert np.abs(conf[0].numpy() - ref).max() < 1e-5 * max(1.0, ref.max() / 1e-2)


def test_dual_softmax_read_out(golden):
    g = golden("3dmatch_branches")
    W, sc = soft_weights(), scene()
    cfg = dict(synth.VARIANTS["3dmatch"], match_type="dual_softmax", dsmax_temperature=float(g["dsm_temperature"]))
    pe_s, pe_t = orc.position_code(cfg, sc["p_s"]), orc.position_code(cfg, sc["p_t"])
    sm_p, tm_p = masks(64, 48, 50, 41)
    for mtag, (sm, tm) in (("mask", (sm_p, tm_p)), ("none", (None, None))):
        conf = orc.match_head(W, cfg, sc["f_s"], sc["f_t"], pe_s, pe_t, sm, tm)
        ref = g["dsm_conf_" + mtag]
        assert conf.shape[1:] == ref.shape and np.abs(conf[0].numpy() - ref).max() < 2e-6
    assert g["dsm_conf_mask"][50:].max() == 0.0 and g["dsm_conf_mask"][:, 41:].max() == 0.0      # both masks zero their rows / columns


@pytest.mark.parametrize("tag", ["sin", "rot_ent"])
def test_loop_of_a_form_matches_the_reference(golden, tag):
    """the reference's own Pipeline.forward with the branch selected, three steps, warp fed back"""
    g = golden("3dmatch_branches")
    N, M, steps, seed = (int(a) for a in g["loop_shape"])
    W, cfg = soft_weights(), form_cfg(tag)
    p = synth.make_pair(N, M, cfg["C"], seed=seed)
    q = lambda k: T(p[k])[None]
    ms, mt = masks(N, M)
    trace = []
    out = orc.denoise_loop(W, cfg, q("src_feats"), q("tgt_feats"), q("s_pcd"), q("t_pcd"), ms, mt, q("x_T"), steps, 200.0, trace=trace)
    for k in range(steps):
        assert np.abs(trace[k]["R_forwd"][0].numpy() - g["loop_%s_R_forwd" % tag][k]).max() < 1e-4, k
        assert np.abs(trace[k]["t_forwd"][0].numpy() - g["loop_%s_t_forwd" % tag][k]).max() < 1e-4, k
        assert abs(float(trace[k]["cond"][0]) - float(g["loop_%s_cond" % tag][k])) < 1e-4 * float(g["loop_%s_cond" % tag][k]), k
        assert np.abs(trace[k]["x0"][0].numpy() - g["loop_%s_x0" % tag][k]).max() < 1e-5, k
    conf = out["conf_matrix_pred"]
    assert conf.dtype == torch.float64 and np.abs(conf[0].numpy() - g["loop_%s_conf" % tag]).max() < 1e-6


def test_4d_loop_of_the_sinusoidal_form_matches_the_reference(golden):
    """the same branch in the 4DMatch tree (no min-shift, sigma * xi, sigmoid read-out, masks; step 0's fit fails the condition gate -> identity warp)"""
    g = golden("4dmatch_branches")
    N, M, nv, mv, steps, seed = (int(a) for a in g["loop_shape"])
    mc = float(g["loop_mc"])
    v = synth.VARIANTS["4dmatch"]
    cfg = dict(v, pe_type="sinusoidal", entangled=False)
    from diffreg_hip.synth import make_weights
    from oracle.make_golden import HEAD_GAIN_SOFT
    W = {k: T(a) for k, a in make_weights(v["C"], seed=7, head_gain=HEAD_GAIN_SOFT).items()}
    p = synth.make_pair(N, M, v["C"], seed=seed)
    q = lambda k: T(p[k])[None]
    ms, mt = masks(N, M, nv, mv)
    noise = T(synth.step_noise(N, M, seed, steps))[:, None]
    trace = []
    out = orc.denoise_loop(W, cfg, q("src_feats"), q("tgt_feats"), q("s_pcd"), q("t_pcd"), ms, mt, q("x_T"), steps, mc, variant="4dmatch", noise=noise,
                           trace=trace)
    for k in range(steps):
        assert np.abs(trace[k]["R_forwd"][0].numpy() - g["loop_sin_R_forwd"][k]).max() < 1e-4, k
        assert abs(float(trace[k]["cond"][0]) - float(g["loop_sin_cond"][k])) < 1e-4 * float(g["loop_sin_cond"][k]), k
        assert np.abs(trace[k]["x0"][0].numpy() - g["loop_sin_x0"][k]).max() < 1e-5, k
    assert np.abs(out["conf_matrix_pred"][0].numpy() - g["loop_sin_conf"]).max() < 1e-5


@pytest.mark.parametrize("form", ["sin", "rot_ent", "sin_ent", "dsm"])
def test_training_gradients_of_a_form_oracle_autograd_against_reference(golden, form):
    """The denoising-branch training graph of a non-default form (oracle.denoiser + match_head + train_oracle.focal_loss, float64 weights, torch
    autograd) against the reference's own float32 backward (oracle/make_golden_train_branches.py): loss, conf, feature gradients and a parameter
    gradient of every layer.  This is the link that pins what the GPU tests of the same vectors compare the device with; the case is off every
    ReLU kink by 5e-6 (see the minting script), so the float64 restatement and the float32 reference are on the same side of all of them."""
    from oracle import train_oracle as tro
    from tests.helpers import train_branch_case
    g = golden("train_backward_branches")
    c = train_branch_case(40, 32, 70)
    scale = float(g[form + "_feat_scale"])
    pe_type, ent, mtype = {"sin": ("sinusoidal", False, "sinkhorn"), "rot_ent": ("rotary", True, "sinkhorn"), "sin_ent": ("sinusoidal", True, "sinkhorn"),
                           "dsm": ("rotary", False, "dual_softmax")}[form]
    v = dict(synth.VARIANTS["3dmatch"], pe_type=pe_type, entangled=ent, match_type=mtype, dsmax_temperature=float(g["dsm_temperature"]))
    W = {k: a.double().clone().requires_grad_(True) for k, a in soft_weights().items() if k.startswith("denoising_")}
    fs = (c["f_s"] * scale).double().requires_grad_(True)
    ft = (c["f_t"] * scale).double().requires_grad_(True)
    hs, ht, pe_s, pe_t = orc.denoiser(W, v, fs, ft, c["warped"], c["p_t"], c["src_mask"], c["tgt_mask"])
    hat = orc.match_head(W, v, hs, ht, pe_s, pe_t, c["src_mask"], c["tgt_mask"])
    gt = torch.zeros_like(hat)
    gt[0][c["matches"][0][0], c["matches"][0][1]] = 1
    loss = tro.focal_loss(hat, gt, match_type=mtype)
    loss.backward()
    pre = form + "_branch_"
    assert abs(float(loss) - float(g[pre + "loss32"])) <= 1e-4 * float(g[pre + "loss32"])
    assert np.abs(hat.detach().numpy() - g[pre + "conf32"]).max() <= 1e-5
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(fs.grad[0, ::3, ::4].numpy(), g[pre + "grad_src32"]) < 1e-3 and rel(ft.grad[0, ::3, ::4].numpy(), g[pre + "grad_tgt32"]) < 1e-3
    st = int(g["stride"])
    for l in range(6):
        for name in ("q_proj.weight", "mlp.0.weight", "norm2.bias"):
            got = W["denoising_transformer.layers.%d.%s" % (l, name)].grad
            got = (got[::st, ::st] if got.dim() == 2 else got).numpy()
            assert rel(got, g["%sg32_layers.%d.%s" % (pre, l, name)]) < 1e-3, (l, name)
    got = W["denoising_coarse_matching.src_proj.weight"].grad[::st, ::st].numpy()
    assert rel(got, g[pre + "g32_head.src_proj.weight"]) < 1e-3
