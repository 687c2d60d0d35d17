"""The configuration branches no shipped yaml selects, through the overlay's modules on the device, against vectors minted by the reference
(oracle/make_golden_branches.py): pe_type 'sinusoidal' (3D/models/position_encoding.py:43-44, 68-69; transformero.py:50-57), entangled = True
(transformero.py:234-254; matching.py:181), match_type 'dual_softmax' (matching.py:193-205), positioning_type 'oracle' / 'randSO3'
(transformero.py:202-216, 261-280), and Pipeline.forward's evaluation loop with such a branch selected (pipeline.py:221-283).  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import lib, synth
from tests.helpers import T, masks, train_weights, guarded
from tests.test_models_api_gpu import ref_like_config, to_attr, StubBackbone
from tests.test_branches_oracle import FORMS, scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def form_config(tag, steps=3, mc=200.0, match_type="sinkhorn"):
    pe_type, ent = FORMS[tag] if tag in FORMS else ("rotary", False)
    cfg = ref_like_config("3dmatch", steps, mc)
    ct = cfg.coarse_transformer
    ct["pe_type"], ct["entangled"] = pe_type, ent
    ct["feature_matching"]["entangled"] = ent
    cfg.coarse_matching["entangled"] = ent
    cfg.coarse_matching["match_type"] = ct["feature_matching"]["match_type"] = match_type
    return cfg


def sub(W, pre, drop=()):
    return {k[len(pre):]: t for k, t in W.items() if k.startswith(pre) and k[len(pre):] not in drop}


def match_sets_equal(got, ref):
    return set(map(tuple, got.cpu().tolist())) == set(map(tuple, np.asarray(ref).tolist()))


def test_sinusoidal_position_code(golden):
    from models.position_encoding import VolumetricPositionEncoding
    g = golden("3dmatch_branches")
    pe = VolumetricPositionEncoding(form_config("sin").coarse_transformer)
    code = pe(scene()["p_s"].to(DEV))
    assert code.shape == (1, 64, synth.VARIANTS["3dmatch"]["C"])
    assert np.abs(code[0, :16].cpu().numpy() - g["pe_sin"]).max() < 2e-6


@pytest.mark.parametrize("tag", list(FORMS))
def test_layer_denoiser_and_head_of_every_form(golden, tag):
    from models.matching import Matching
    from models.position_encoding import VolumetricPositionEncoding
    from models.transformero import RepositioningTransformer
    g = golden("3dmatch_branches")
    W, sc, cfg = train_weights("soft"), scene(), form_config(tag)
    pe_type, ent = FORMS[tag]
    ct = to_attr(dict(cfg.coarse_transformer))
    ct["layer_types"] = list(synth.LAYER_TYPES)
    den = RepositioningTransformer(ct)
    den.load_state_dict(sub(W, "denoising_transformer."))
    head = Matching(cfg.coarse_matching)
    head.load_state_dict(sub(W, "denoising_coarse_matching."))
    den, head = den.to(DEV).eval(), head.to(DEV).eval()
    d = lambda k: sc[k].to(DEV)
    sm_p, tm_p = (m.to(DEV) for m in masks(64, 48, 50, 41))
    sm_f, tm_f = (m.to(DEV) for m in masks(64, 48))
    pe_mod = VolumetricPositionEncoding(ct)
    pes, pet = (None, None) if ent else (pe_mod(d("p_s")), pe_mod(d("p_t")))
    lay = den.layers[1]
    out = lay(d("f_s"), d("f_t"), pes, pet, sm_p, tm_p)
    assert np.abs(out[0].cpu().numpy() - g[tag + "_layer_cross_mask"]).max() < 1e-4
    out = lay(d("f_s"), d("f_s"), pes, pes, sm_f, sm_f)
    assert np.abs(out[0].cpu().numpy() - g[tag + "_layer_self_full"]).max() < 1e-4
    for mtag, (sm, tm) in (("full", (sm_f, tm_f)), ("mask", (sm_p, tm_p))):
        data = {}
        hs, ht, pe_s, pe_t = den(d("f_s"), d("f_t"), d("p_s"), d("p_t"), sm, tm, data)
        assert np.abs(hs[0].cpu().numpy() - g["%s_den_fs_%s" % (tag, mtag)]).max() < 1e-4
        assert np.abs(ht[0].cpu().numpy() - g["%s_den_ft_%s" % (tag, mtag)]).max() < 1e-4
        conf, match = head(hs, ht, pe_s, pe_t, sm, tm, data, pe_type=pe_type)
        ref = g["%s_conf_%s" % (tag, mtag)]
        assert np.abs(conf[0].cpu().numpy() - ref).max() < 1e-4 * max(ref.max(), 1e-2)      # (confidences of 1e-2: 1e-4 RELATIVE to the largest)
        assert np.abs(data["src_feats"][0, :8].cpu().numpy() - g["%s_feats_pos_%s" % (tag, mtag)]).max() < 1e-3
        assert match.shape == g["%s_match_%s" % (tag, mtag)].shape and match_sets_equal(match, g["%s_match_%s" % (tag, mtag)])


def test_dual_softmax_read_out(golden):
    from models.matching import Matching
    from models.position_encoding import VolumetricPositionEncoding
    g = golden("3dmatch_branches")
    W, sc = train_weights("soft"), scene()
    cfg = form_config("rotary", match_type="dual_softmax")
    head = Matching(cfg.coarse_matching)
    assert not hasattr(head, "bin_score")                                  # matching.py:113-121: that branch creates none
    head.load_state_dict(sub(W, "denoising_coarse_matching.", drop=("bin_score",)))
    head = head.to(DEV).eval()
    pe_mod = VolumetricPositionEncoding(cfg.coarse_transformer)
    d = lambda k: sc[k].to(DEV)
    pes, pet = pe_mod(d("p_s")), pe_mod(d("p_t"))
    sm_p, tm_p = (m.to(DEV) for m in masks(64, 48, 50, 41))
    for mtag, (sm, tm) in (("mask", (sm_p, tm_p)), ("none", (None, None))):
        conf, match = head(d("f_s"), d("f_t"), pes, pet, sm, tm, {}, pe_type="rotary")
        ref = g["dsm_conf_" + mtag]
        assert np.abs(conf[0].cpu().numpy() - ref).max() < 2e-6
        assert match_sets_equal(match, g["dsm_match_" + mtag])
    # the kernel on a ragged batch between guard bands, against torch's own softmaxes (sharp logits: temperature 0.1)
    P, N, M = 3, 150, 333
    sim = T(synth.hash_normal(3, 91, (P, N, M)).astype(np.float32)).to(DEV)
    sm = (torch.arange(N)[None] < torch.tensor([[150], [97], [1]])).to(DEV)
    tm = (torch.arange(M)[None] < torch.tensor([[333], [200], [333]])).to(DEV)
    s1 = (sim / 0.1).masked_fill(~sm[:, :, None], float("-inf"))
    s2 = (sim / 0.1).masked_fill(~tm[:, None, :], float("-inf"))
    ref = (torch.softmax(s1.double(), 1) * torch.softmax(s2.double(), 2)).cpu()
    got = lib.dual_softmax(sim, 0.1, sm, tm).cpu()
    assert torch.isfinite(got).all() and (got.double() - ref).abs().max().item() < 2e-6
    assert got[1, 97:].abs().max().item() == 0.0 and got[1, :, 200:].abs().max().item() == 0.0


def test_oracle_and_random_positioning_layers(golden):
    from models.transformero import RepositioningTransformer
    g = golden("3dmatch_branches")
    W, sc = train_weights("soft"), scene()
    ct = to_attr(dict(form_config("rotary").coarse_transformer))
    ct["positioning_type"] = "oracle"
    cot = RepositioningTransformer(ct)
    cot.load_state_dict({k: t for k, t in sub(W, "coarse_transformer.").items() if not k.startswith("layers.2.")})
    cot = cot.to(DEV).eval()
    d = lambda k: sc[k].to(DEV)
    sm_p, tm_p = (m.to(DEV) for m in masks(64, 48, 50, 41))
    data = {"batched_rot": sc["R_gt"][None].to(DEV), "batched_trn": sc["t_gt"].view(1, 3, 1).to(DEV)}
    hs, ht, pe_s, pe_t = cot(d("f_s"), d("f_t"), d("p_s"), d("p_t"), sm_p, tm_p, data)
    assert np.abs(hs[0].cpu().numpy() - g["oracle_pos_fs"]).max() < 1e-4 and np.abs(ht[0].cpu().numpy() - g["oracle_pos_ft"]).max() < 1e-4
    assert np.abs(pe_s[0, :8].cpu().numpy() - g["oracle_pos_pe_s"]).max() < 2e-5          # (the code of the RE-POSED source)
    np.random.seed(5)
    moved = cot.rand_rot_pcd(d("p_s").clone(), sm_p)
    assert np.abs(moved[0].cpu().numpy() - g["rand_rot_pcd"]).max() < 1e-5
    ct["positioning_type"] = "randSO3"
    rnd = RepositioningTransformer(ct)
    rnd.load_state_dict({k: t for k, t in sub(W, "coarse_transformer.").items() if not k.startswith("layers.2.")})
    hs2, _, _, _ = rnd.to(DEV).eval()(d("f_s"), d("f_t"), d("p_s").clone(), d("p_t"), sm_p, tm_p, {})
    assert torch.isfinite(hs2[:, :50]).all()


@pytest.mark.parametrize("tag", ["sin", "rot_ent"])
def test_pipeline_evaluation_loop_of_a_form(golden, tag):
    """Pipeline.forward with the branch selected: the module-level loop (no fused engine), three steps, warp fed back; the smallest K-th gap of
    the three selections is 2.8e-4 relative in the oracle's run, so the free-running trajectory is comparable"""
    from models.pipeline import Pipeline
    g = golden("3dmatch_branches")
    N, M, steps, seed = (int(a) for a in g["loop_shape"])
    cfg = form_config(tag, steps=steps)
    model = Pipeline(cfg, backbone=StubBackbone())
    assert not model._fused_loop_config()
    W = train_weights("soft")
    sd = model.state_dict()
    sd.update({k: t for k, t in W.items() if k in sd})
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    C = synth.VARIANTS["3dmatch"]["C"]
    p = synth.make_pair(N, M, C, seed=seed)
    feats = torch.cat([T(p["src_feats"]), T(p["tgt_feats"])]).to(DEV)
    pts = torch.cat([T(p["s_pcd"]), T(p["t_pcd"])]).to(DEV)
    ms, mt = masks(N, M)
    data = {"points": [None, None, pts, None], "src_mask": ms.to(DEV), "tgt_mask": mt.to(DEV), "_feats": feats,
            "src_ind_coarse_split": torch.arange(N, device=DEV), "tgt_ind_coarse_split": torch.arange(M, device=DEV),
            "src_ind_coarse": torch.arange(N, device=DEV), "tgt_ind_coarse": torch.arange(N, N + M, device=DEV), "x_T": T(p["x_T"])[None].to(DEV)}
    x0_log, warp_log = [], []
    head_fwd, proc_fwd = model.denoising_coarse_matching.forward, model.denoising_soft_procrustes.forward

    def head_spy(*a, **k):
        r = head_fwd(*a, **k)
        x0_log.append(r[0].detach().clone())
        return r

    def proc_spy(*a, **k):
        r = proc_fwd(*a, **k)
        warp_log.append([z.detach().clone() for z in r])
        return r
    model.denoising_coarse_matching.forward, model.denoising_soft_procrustes.forward = head_spy, proc_spy
    out = model(data)
    assert len(x0_log) == steps and len(warp_log) == steps
    worst = 0.0
    for k in range(steps):
        assert np.abs(warp_log[k][2][0].cpu().numpy() - g["loop_%s_R_forwd" % tag][k]).max() < 1e-4, k
        assert np.abs(warp_log[k][3][0].cpu().numpy() - g["loop_%s_t_forwd" % tag][k]).max() < 1e-4, k
        assert abs(float(warp_log[k][4][0]) - float(g["loop_%s_cond" % tag][k])) < 1e-4 * float(g["loop_%s_cond" % tag][k]), k
        e = np.abs(x0_log[k][0].cpu().numpy() - g["loop_%s_x0" % tag][k]).max()
        worst = max(worst, e)
        assert e < 1e-4 * max(g["loop_%s_x0" % tag][k].max(), 1e-2), (k, e)
    conf = out["conf_matrix_pred"]
    ref = g["loop_%s_conf" % tag]
    assert conf.dtype == torch.float64 and np.abs(conf[0].cpu().numpy() - ref).max() < 1e-4 * ref.max()
    assert match_sets_equal(out["match_pred"], g["loop_%s_match_pred" % tag])
    assert out["R_s2t_pred"].shape == (1, 3, 3)
    print("branch loop %s: worst x_start deviation %.3g, conf %.3g (largest conf %.3g)" % (tag, worst, np.abs(conf[0].cpu().numpy() - ref).max(), ref.max()))


def test_pipeline_evaluation_loop_of_a_form_in_the_4d_variant(golden):
    """the module-level loop's 4DMatch branch (4D/models/pipeline.py:155-197: no min-shift, sigma * xi from data['noise'], sigmoid read-out, masks)
    with pe_type 'sinusoidal', against the reference's own Pipeline.forward in the 4DMatch tree"""
    from models.pipeline import Pipeline
    from oracle.make_golden import HEAD_GAIN_SOFT
    g = golden("4dmatch_branches")
    N, M, nv, mv, steps, seed = (int(a) for a in g["loop_shape"])
    mc = float(g["loop_mc"])
    v = synth.VARIANTS["4dmatch"]
    cfg = ref_like_config("4dmatch", steps, mc)
    cfg.coarse_transformer["pe_type"] = "sinusoidal"
    model = Pipeline(cfg, backbone=StubBackbone())
    assert model.variant == "4dmatch" and not model._fused_loop_config()
    W = {k: T(a) for k, a in synth.make_weights(v["C"], seed=7, head_gain=HEAD_GAIN_SOFT).items()}
    sd = model.state_dict()
    sd.update({k: t for k, t in W.items() if k in sd})
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    p = synth.make_pair(N, M, v["C"], seed=seed)
    feats = torch.cat([T(p["src_feats"]), T(p["tgt_feats"])]).to(DEV)
    pts = torch.cat([T(p["s_pcd"]), T(p["t_pcd"])]).to(DEV)
    ms, mt = masks(N, M, nv, mv)
    data = {"points": [None, None, pts, None], "src_mask": ms.to(DEV), "tgt_mask": mt.to(DEV), "_feats": feats,
            "src_ind_coarse_split": torch.arange(N, device=DEV), "tgt_ind_coarse_split": torch.arange(M, device=DEV),
            "src_ind_coarse": torch.arange(N, device=DEV), "tgt_ind_coarse": torch.arange(N, N + M, device=DEV), "x_T": T(p["x_T"])[None].to(DEV),
            "noise": T(synth.step_noise(N, M, seed, steps))[:, None].to(DEV)}
    x0_log, warp_log = [], []
    head_fwd, proc_fwd = model.denoising_coarse_matching.forward, model.denoising_soft_procrustes.forward

    def head_spy(*a, **k):
        r = head_fwd(*a, **k)
        x0_log.append(r[0].detach().clone())
        return r

    def proc_spy(*a, **k):
        r = proc_fwd(*a, **k)
        warp_log.append([z.detach().clone() for z in r])
        return r
    model.denoising_coarse_matching.forward, model.denoising_soft_procrustes.forward = head_spy, proc_spy
    out = model(data)
    for k in range(steps):
        assert np.abs(warp_log[k][2][0].cpu().numpy() - g["loop_sin_R_forwd"][k]).max() < 1e-4, k
        assert np.abs(warp_log[k][3][0].cpu().numpy() - g["loop_sin_t_forwd"][k]).max() < 1e-4, k
        assert abs(float(warp_log[k][4][0]) - float(g["loop_sin_cond"][k])) < 1e-4 * float(g["loop_sin_cond"][k]), k
        assert np.abs(x0_log[k][0].cpu().numpy() - g["loop_sin_x0"][k]).max() < 1e-4, k
    conf = out["conf_matrix_pred"]
    assert conf.dtype == torch.float64 and np.abs(conf[0].cpu().numpy() - g["loop_sin_conf"]).max() < 1e-4
    assert "match_pred" not in out                                   # (the 4D tree has no read-out list in Pipeline.forward)


def test_module_loop_pairs_of_a_batch_do_not_see_each_other():
    """the module-level loop on a batch of two different pairs = the two pairs run alone: the 3DMatch min-shift is taken per PAIR (the reference only
    ever holds a 1 x N x M state, 3D/models/pipeline.py:238; with the learned bin score the Sinkhorn result is not shift-invariant, so a batch-wide
    minimum would make a pair's result depend on its neighbours)"""
    from models.pipeline import Pipeline
    N, M, steps = 96, 80, 3
    cfg = form_config("sin", steps=steps)
    model = Pipeline(cfg, backbone=StubBackbone())
    W = train_weights("soft")
    sd = model.state_dict()
    sd.update({k: t for k, t in W.items() if k in sd})
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    C = synth.VARIANTS["3dmatch"]["C"]
    prs = [synth.make_pair(N, M, C, seed=410 + i) for i in range(2)]
    prs[1]["x_T"] = prs[1]["x_T"] - 3.0                      # the second pair's state sits lower: a batch-wide minimum would shift the first pair by ~3
    st = lambda k, sel: torch.from_numpy(np.stack([prs[i][k] for i in sel])).to(DEV)

    def run(sel):
        ms, mt = torch.ones(len(sel), N, dtype=torch.bool, device=DEV), torch.ones(len(sel), M, dtype=torch.bool, device=DEV)
        data = {"x_T": st("x_T", sel)}
        with torch.no_grad():
            model._eval_loop_on_modules(data, st("src_feats", sel), st("tgt_feats", sel), st("s_pcd", sel), st("t_pcd", sel), ms, mt)
        return data["conf_matrix_pred"]
    both, alone = run([0, 1]), [run([0]), run([1])]
    for i in range(2):
        err = float((both[i] - alone[i][0]).abs().max())
        assert err < 1e-5 * float(alone[i][0].max()), (i, err, float(alone[i][0].max()))
