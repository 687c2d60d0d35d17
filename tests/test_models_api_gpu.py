"""The drop-in `models` package: same names/signatures as the reference, HIP underneath.  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, weights, pair, masks, sinkhorn_case, guarded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def to_attr(d):
    return AttrDict({k: to_attr(v) for k, v in d.items()}) if isinstance(d, dict) else d


def ref_like_config(variant, steps, mc):
    v = synth.VARIANTS[variant]
    matching = dict(feature_dim=v["C"], confidence_threshold=0.2, entangled=False, dsmax_temperature=0.1,
                    match_type="sinkhorn", skh_init_bin_score=1.0, skh_iters=3, skh_prefilter=False)
    return to_attr(dict(dataset=variant, kpfcn_config=dict(coarse_level=-2, coarse_feature_dim=v["C"]), coarse_matching=matching,
                        coarse_transformer=dict(feature_dim=v["C"], n_head=v["H"],
                                                layer_types=["self", "cross", "positioning", "self", "cross"],
                                                positioning_type="procrustes", pe_type="rotary",
                                                vol_bnds=[list(v["origin"]), [1.093, 0.78, 2.92]], voxel_size=v["voxel"],
                                                feature_matching=dict(matching), entangled=False,
                                                procrustes=dict(max_condition_num=mc, sample_rate=1.0)),
                        SAMPLE_STEP=steps))


class StubBackbone(torch.nn.Module):
    def forward(self, data, phase="coarse"):
        return data["_feats"]


def test_log_optimal_transport_and_modules():
    from models.matching import log_optimal_transport, Matching
    from models.position_encoding import VolumetricPositionEncoding
    from models.procrustes import SoftProcrustesLayer
    from models.transformero import GeometryAttentionLayer
    sc, sm, tm = sinkhorn_case(96, 80, 70, 61, torch.float32)
    Z = log_optimal_transport(sc.to(DEV), torch.tensor(1.0, device=DEV), 3, sm.to(DEV), tm.to(DEV)).cpu()
    ref = orc.sinkhorn_log(sc, torch.tensor(1.0), 3, sm, tm)
    fin = torch.isfinite(ref)
    assert (Z[fin] - ref[fin]).abs().max().item() < 1e-4
    variant = "3dmatch"
    v = synth.VARIANTS[variant]
    cfg = ref_like_config(variant, 1, 200)
    _, p = pair(variant, 64, 48, 3)
    pe_mod = VolumetricPositionEncoding(cfg.coarse_transformer)
    code = pe_mod(p["p_s"].to(DEV))
    cos, sin = orc.vol_pe(p["p_s"], v["C"], v["origin"], v["voxel"])
    assert code.shape == (1, 64, v["C"], 2)
    assert (code[..., 0].cpu() - cos).abs().max().item() < 2e-6 and (code[..., 1].cpu() - sin).abs().max().item() < 2e-6
    W = weights(variant)
    layer = GeometryAttentionLayer(cfg.coarse_transformer)
    pre = "denoising_transformer.layers.1."
    layer.load_state_dict({k[len(pre):]: t for k, t in W.items() if k.startswith(pre)})
    layer = layer.to(DEV).eval()
    code_t = pe_mod(p["p_t"].to(DEV))
    out = layer(p["f_s"].to(DEV), p["f_t"].to(DEV), code, code_t, None, None).cpu()
    ref_l = orc.attention_layer(W, pre, p["f_s"], p["f_t"], (cos, sin), orc.vol_pe(p["p_t"], v["C"], v["origin"], v["voxel"]),
                                None, None, v["H"])
    assert (out - ref_l).abs().max().item() < 1e-4
    head = Matching(cfg.coarse_matching)
    head.load_state_dict({k[len("denoising_coarse_matching."):]: t for k, t in W.items() if k.startswith("denoising_coarse_matching.")})
    head = head.to(DEV).eval()
    ms, mt = masks(64, 48)
    data = {}
    conf, cm = head(p["f_s"].to(DEV), p["f_t"].to(DEV), code, code_t, ms.to(DEV), mt.to(DEV), data, pe_type="rotary")
    ref_c = orc.match_head(W, v, p["f_s"], p["f_t"], (cos, sin), orc.vol_pe(p["p_t"], v["C"], v["origin"], v["voxel"]), ms, mt)
    assert (conf.cpu() - ref_c).abs().max().item() < 1e-4 and "src_feats_nopos" in data and cm.shape[1] == 3
    proc = SoftProcrustesLayer(cfg.coarse_transformer.procrustes)
    r = proc(ref_c.to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), ms.to(DEV), mt.to(DEV))
    r_ref = orc.procrustes(ref_c, p["p_s"], p["p_t"], ms, mt, 1.0, 200)
    assert (r[0].cpu() - r_ref[0]).abs().max().item() < 1e-4 and (r[1].cpu() - r_ref[1]).abs().max().item() < 1e-4
    # float64 conf: default = well-defined (R,t); strict_reference = the reference's identity fallback (Q3)
    r64 = proc(ref_c.double().to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), ms.to(DEV), mt.to(DEV), strict_reference=True)
    assert torch.equal(r64[0].cpu(), torch.eye(3, dtype=torch.float64)[None])


@pytest.mark.parametrize("variant,N,M,nv,mv,steps,mc,seed,tag", [
    ("3dmatch", 128, 128, 128, 128, 1, 200, 11, "n128_s1_mc200"),
    ("4dmatch", 128, 128, 112, 100, 5, 40, 21, "n128_s5_mc40_masked")])
def test_pipeline_forward_matches_reference(golden, variant, N, M, nv, mv, steps, mc, seed, tag):
    """Pipeline(config).forward(data) with the reference's input dict and state-dict layout."""
    from models.pipeline import Pipeline
    g = golden("%s_loop_%s" % (variant, tag))
    cfg = ref_like_config(variant, steps, mc)
    model = Pipeline(cfg, backbone=StubBackbone())
    assert cfg.coarse_transformer.layer_types == ["self", "cross"] * 3            # config mutation (Q12)
    sd = model.state_dict()
    assert sd["alphas_cumprod"].dtype == torch.float64 and "denoising_coarse_matching.tgt_proj.weight" in sd
    W = weights(variant)
    missing = [k for k in W if k not in sd]
    assert not missing
    sd.update({("module." + k)[7:]: t for k, t in W.items()})
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    pr, p = pair(variant, N, M, seed)
    feats = torch.cat([p["f_s"][0], p["f_t"][0]]).to(DEV)
    pts = torch.cat([p["p_s"][0], p["p_t"][0]]).to(DEV)
    ms, mt = masks(N, M, nv, mv)
    data = {"points": [None, None, pts, None], "src_mask": ms.to(DEV), "tgt_mask": mt.to(DEV), "_feats": feats,
            "src_ind_coarse_split": torch.arange(N, device=DEV), "tgt_ind_coarse_split": torch.arange(M, device=DEV),
            "src_ind_coarse": torch.arange(N, device=DEV), "tgt_ind_coarse": torch.arange(N, N + M, device=DEV),
            "x_T": p["x_T"].to(DEV), "noise": T(synth.step_noise(N, M, seed, steps))[:, None].to(DEV)}
    out = model(data)
    conf = out["conf_matrix_pred"]
    assert conf.dtype == torch.float64 and conf.shape == (1, N, M)
    # the loop fixture's rule (tests/test_loop_gpu.py): a plain 1e-4 on every entry but the committed ill-conditioned ones, and on those at
    # least as close to the float64 evaluation as twice the reference's own float32 run
    from tests.test_loop_gpu import assert_matrix_parity, exemptions, f64_evaluation
    _, conf_f64 = f64_evaluation(variant, N, M, nv, mv, steps, mc, seed)
    assert_matrix_parity(conf[0].cpu().numpy(), g["conf"], conf_f64, "Pipeline.forward conf", exemptions("%s_loop_%s" % (variant, tag), "conf"))
    assert out["R_s2t_pred"].shape == (1, 3, 3) and out["t_s2t_pred"].shape == (1, 3, 1)
    assert torch.equal(out["s_pcd"].cpu(), p["p_s"])
    if variant == "3dmatch":
        mp = out["match_pred"]
        assert mp.dtype == torch.int64 and mp.shape[1] == 3 and N <= mp.shape[0] <= N + M
        # the final (R,t): the reference returns identity (Q3); ours is the fit on float32(conf)
        r = orc.procrustes(conf.cpu().float(), p["p_s"], p["p_t"], ms, mt, 1.0, mc)
        assert (out["R_s2t_pred"].cpu() - r[0]).abs().max().item() < 1e-4
        model.strict_reference = True
        out2 = model(dict(data))
        assert torch.equal(out2["R_s2t_pred"].cpu().float(), torch.from_numpy(g["R_s2t_pred"]).float())
    # side effects of the eval branch on `data`: the last Matching.forward leaves src_proj(feats) with and without the
    # rotary embedding (matching.py:177-187), the denoiser an empty position_layers dict (transformero.py:172)
    assert out["position_layers"] == {}
    tr = []
    v = synth.VARIANTS[variant]
    orc.denoise_loop(W, v, p["f_s"], p["f_t"], p["p_s"], p["p_t"], ms, mt, p["x_T"], steps, mc, variant=variant,
                     noise=T(synth.step_noise(N, M, seed, steps))[:, None], trace=tr)
    hs, ht, pe_s, pe_t = orc.denoiser(W, v, p["f_s"], p["f_t"], tr[-1]["warped"], p["p_t"].float(), ms, mt)
    Wp = W["denoising_coarse_matching.src_proj.weight"]
    for key, ref_t in (("src_feats_nopos", hs @ Wp.T), ("tgt_feats_nopos", ht @ Wp.T), ("src_feats", orc.rotary(hs @ Wp.T, *pe_s)),
                       ("tgt_feats", orc.rotary(ht @ Wp.T, *pe_t))):
        got = out[key].cpu()
        assert got.shape == ref_t.shape
        # (features of the LAST step: they carry the loop's accumulated float32 deviation in the warp -- measured 2e-5 in R after
        #  5 steps, which is 1.4e-3 rad in the highest-frequency rotary angle at voxel 0.04 -- times the head gain of the
        #  synthetic weights: 1e-3 of the largest entry without the position code, 1e-2 with it)
        tol = (1e-3 if key.endswith("nopos") else 1e-2) * max(1.0, ref_t.abs().max().item())
        assert (got - ref_t).abs().max().item() < tol, key
    # eval_flag=True (validation) skips the loop (pipeline.py:221)
    d2 = dict(data); d2.pop("conf_matrix_pred", None)
    assert "conf_matrix_pred" not in model(d2, eval_flag=True)


def test_pipeline_end_to_end_with_overlay_backbone(golden):
    """Rows f1 + a1: Pipeline(config) with its OWN backbone (models.backbone.KPFCN of the overlay) from the collate-style
    input dict (stacked points, neighbour / pool / upsample indices) to conf_matrix_pred, against the oracle backbone
    followed by the oracle loop."""
    from models.pipeline import Pipeline
    from oracle import kpfcn_oracle as ko
    from tests.test_oracle_golden import kpfcn_inputs
    variant, steps, mc = "3dmatch", 2, 200
    g, bsd, tb = kpfcn_inputs(golden)
    cfg = ref_like_config(variant, steps, mc)
    cfg.kpfcn_config = to_attr(dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), KP_influence="linear", aggregation_mode="sum",
                                    deformable=False, use_batch_norm=True, fine_feature_dim=264, coarse_level=-2))
    model = Pipeline(cfg)
    assert type(model.backbone).__module__.endswith("backbone") and hasattr(model.backbone, "encoder_blocks")
    sd = model.state_dict()
    W = weights(variant)
    sd.update(W)
    sd.update({"backbone." + k: v for k, v in bsd.items()})
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    b = synth.make_kpfcn_batch()
    ns, nt = b["stack_lengths"][2]                                   # coarse_level = -2 -> layer 2
    x_T = T(synth.hash_normal(77, 5, (1, ns, nt))).float()
    data = {k: [t.to(DEV) for t in v] for k, v in tb.items() if isinstance(v, list)}
    data["features"] = tb["features"].to(DEV)
    data.update({"src_mask": torch.ones(1, ns, dtype=torch.bool, device=DEV), "tgt_mask": torch.ones(1, nt, dtype=torch.bool, device=DEV),
                 "src_ind_coarse_split": torch.arange(ns, device=DEV), "tgt_ind_coarse_split": torch.arange(nt, device=DEV),
                 "src_ind_coarse": torch.arange(ns, device=DEV), "tgt_ind_coarse": torch.arange(ns, ns + nt, device=DEV), "x_T": x_T.to(DEV)})
    out = model(data)
    conf = out["conf_matrix_pred"][0].cpu()
    # oracle: backbone -> split -> loop
    feats = ko.kpfcn_coarse(bsd, tb)
    pts = tb["points"][2]
    v = synth.VARIANTS[variant]
    ms, mt = torch.ones(1, ns, dtype=torch.bool), torch.ones(1, nt, dtype=torch.bool)
    ref = orc.denoise_loop(W, v, feats[None, :ns], feats[None, ns:], pts[None, :ns], pts[None, ns:], ms, mt, x_T, steps, mc, variant=variant)
    # (no percentile: the plain bound + the ill-conditioning rule against a float64 run of the same oracle chain)
    from tests.test_loop_gpu import assert_matrix_parity
    W64 = {k: t.double() for k, t in W.items()}
    f64 = orc.denoise_loop(W64, v, feats[None, :ns].double(), feats[None, ns:].double(), pts[None, :ns], pts[None, ns:], ms, mt, x_T.double(), steps, mc,
                           variant=variant)
    assert_matrix_parity(conf.numpy(), ref["conf_matrix_pred"][0].numpy(), f64["conf_matrix_pred"][0].numpy(), "KPFCN + loop conf")


def test_get_match_on_device_equals_reference_rule():
    """Matching.get_match(conf, thr, mutual=True) -- the 4DMatch tester's read-out of conf_matrix_pred (4D/lib/tester.py:266) --
    through dr_mutual_match_f64 / _f32: index list in nonzero() order, confidences and mask equal to the torch statement of
    3D/models/matching.py:126-143 (oracle.mutual_match), ties included."""
    from models.matching import Matching
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float64, torch.float32):
        conf = torch.rand(3, 70, 45, generator=g, dtype=dtype)
        conf[0, 5, 7] = conf[0, 5, 9] = 2.0                    # a tie of two row maxima that are also column maxima
        conf[1] = torch.sigmoid(8 * (conf[1] - 0.5))
        # NaN (the 3D padded-batch quirk Q8 / Q19): torch.max propagates it, so a row / column holding one has no mutual match
        conf[2, 11, :] = float("nan"); conf[2, 30, 4] = float("nan"); conf[2, 40, 44] = float("nan")
        for thr, mutual in ((0.55, True), (0.0, True), (0.9, False)):
            idx, mc, mask = Matching.get_match(conf.to(DEV), thr, mutual)
            m_ref = conf > thr
            if mutual:
                m_ref = m_ref & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
            i_ref = m_ref.nonzero()
            assert torch.equal(idx.cpu(), i_ref) and torch.equal(mask.cpu(), m_ref)
            assert torch.equal(mc.cpu(), conf[i_ref[:, 0], i_ref[:, 1], i_ref[:, 2]])


def test_scatter_rows_is_split_feats():
    from diffreg_hip import lib
    g = torch.Generator().manual_seed(3)
    src = torch.randn(50, 432, generator=g)
    si = torch.randperm(50, generator=g)[:37]
    di = torch.randperm(64, generator=g)[:37]
    dst = torch.zeros(64, 432, device=DEV)
    lib.scatter_rows(src.to(DEV), si.to(DEV), di.to(DEV), dst)
    ref = torch.zeros(64, 432)
    ref[di] = src[si]
    assert torch.equal(dst.cpu(), ref)
    # ADVICE (round 2): host index tensors are accepted like torch indexing accepts them; negative indices wrap; an out-of-range
    # index raises IndexError (torch's behaviour) and touches nothing outside the buffers
    dst2 = torch.zeros(64, 432, device=DEV)
    lib.scatter_rows(src.to(DEV), si, di, dst2)
    assert torch.equal(dst2.cpu(), ref)
    neg_s, neg_d = torch.tensor([-1, 0, -50]), torch.tensor([-64, 5, -1])
    d4 = torch.zeros(64, 432, device=DEV)
    lib.scatter_rows(src.to(DEV), neg_s, neg_d, d4)
    r4 = torch.zeros(64, 432)
    r4[neg_d] = src[neg_s]
    assert torch.equal(d4.cpu(), r4)
    for bad_s, bad_d in (([50], [0]), ([0], [64]), ([-51], [0]), ([0], [-65])):
        d5, chk = guarded((64, 432), torch.float32, DEV, fill=0.0)
        with pytest.raises(IndexError):
            lib.scatter_rows(src.to(DEV), torch.tensor(bad_s), torch.tensor(bad_d), d5)
        assert not d5.any()
        chk()


def test_engine_cache_is_bounded_and_results_are_copies():
    """ADVICE (round 1): the per-shape cache of DenoiseEngine.run is an LRU of a few entries, a graph is captured only when a
    shape repeats, and run() returns copies (a later run of the same shape does not overwrite results the caller holds)."""
    from diffreg_hip.engine import DenoiseEngine
    variant = "3dmatch"
    v = synth.VARIANTS[variant]
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=1,
                        sk_iters=v["skh_iters"], sample_rate=v["sample_rate"], max_condition_num=200, n_layers=v["n_layers"], device=DEV,
                        cache_entries=2)
    outs = {}
    for n in (32, 40, 48, 32, 32):
        _, p = pair(variant, n, n, 7)
        o = eng.run(p["f_s"].to(DEV), p["f_t"].to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), p["x_T"].to(DEV), graph=True)
        outs.setdefault(n, []).append(o["conf_matrix_pred"])
        assert len(eng._graphs) <= 2
    key32 = [k for k in eng._graphs if k[1] == 32][0]
    assert eng._graphs[key32]["g"] is not None and eng._graphs[key32]["uses"] == 2      # evicted once, seen twice since: captured
    a, b, c = outs[32]
    assert torch.equal(a, b) and torch.equal(b, c) and a.data_ptr() != b.data_ptr()
    _, p2 = pair(variant, 32, 32, 8)
    held = c.clone()
    eng.run(p2["f_s"].to(DEV), p2["f_t"].to(DEV), p2["p_s"].to(DEV), p2["p_t"].to(DEV), p2["x_T"].to(DEV), graph=True)
    assert torch.equal(c, held)                                                          # not overwritten
