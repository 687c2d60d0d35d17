"""Mint golden vectors by RUNNING THE REFERENCE (build container only; needs /root/reference).

    python oracle/make_golden.py            # writes tests/golden/*.npz for both trees

The reference Python is imported from where it lies (never copied); the only shims are the ones
SURVEY.md section 8c lists: a MagicMock for open3d, an attribute-dict in place of EasyDict, a no-op
Tensor.cuda (mutual_topk_select hard-codes .cuda()), and a stub backbone returning the synthetic
coarse features.  Inputs and weights come from diffreg_hip.synth (integer hash), so only the
reference OUTPUTS are stored.  Each tree runs in its own subprocess (both call their package
`models`).
"""
import os
import subprocess
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
TREES = {"3dmatch": "Diff-Reg-3dmatch", "4dmatch": "Diff-Reg-4dmatch"}
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    return d


def ref_config(variant, steps, max_cond):
    from diffreg_hip.synth import VARIANTS
    v = VARIANTS[variant]
    matching = dict(feature_dim=v["C"], confidence_threshold=0.2, entangled=False,
                    dsmax_temperature=0.1, match_type="sinkhorn", skh_init_bin_score=1.0,
                    skh_iters=3, skh_prefilter=False)
    return to_attr(dict(
        kpfcn_config=dict(num_layers=4, in_points_dim=3, first_feats_dim=256, final_feats_dim=32,
                          first_subsampling_dl=0.025, in_feats_dim=1, conv_radius=2.5,
                          deform_radius=5.0, num_kernel_points=15, KP_extent=2.0,
                          KP_influence="linear", aggregation_mode="sum",
                          fixed_kernel_points="center", use_batch_norm=True,
                          batch_norm_momentum=0.02, deformable=False, modulated=False,
                          add_cross_score=True, condition_feature=True,
                          coarse_feature_dim=v["C"], fine_feature_dim=264,
                          coarse_match_radius=0.06, coarse_level=-2),
        coarse_matching=matching,
        coarse_transformer=dict(feature_dim=v["C"], n_head=v["H"],
                                layer_types=["self", "cross", "positioning", "self", "cross"],
                                positioning_type="procrustes", pe_type="rotary",
                                vol_bnds=[list(v["origin"]), [1.093, 0.78, 2.92]],
                                voxel_size=v["voxel"], feature_matching=dict(matching),
                                entangled=False,
                                procrustes=dict(max_condition_num=max_cond, sample_rate=1.0)),
        SAMPLE_STEP=steps))


def run_tree(variant, family="main"):
    import torch
    sys.modules["open3d"] = MagicMock()
    for m in ("easydict", "tensorboardX", "nibabel", "cv2"):
        sys.modules.setdefault(m, MagicMock())
    torch.Tensor.cuda = lambda self, *a, **k: self
    tree = os.path.join(REF, TREES[variant])
    os.chdir(tree)
    sys.path.insert(0, tree)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    from diffreg_hip import synth
    from models.matching import Matching, log_optimal_transport
    from models.position_encoding import VolumetricPositionEncoding
    from models.procrustes import SoftProcrustesLayer
    if variant == "3dmatch":
        from models.transformero import GeometryAttentionLayer, RepositioningTransformer
    else:
        from models.transformer import GeometryAttentionLayer, RepositioningTransformer

    v = synth.VARIANTS[variant]
    C, H = v["C"], v["H"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    os.makedirs(OUT, exist_ok=True)
    save = lambda name, **kw: np.savez_compressed(os.path.join(OUT, "%s_%s.npz" % (variant, name)), **kw)

    def components():
        """F1 .. F6: the ops one by one (main family only)"""
        # ---------------- F1: log_optimal_transport -------------------------------------------
        if variant == "3dmatch":
            f1 = {}
            for (N, M, nv, mv, alpha, dt) in [(128, 128, 128, 128, 1.0, "f32"), (128, 128, 128, 128, 1.0, "f64"),
                                              (200, 256, 200, 256, 0.37, "f32"), (96, 80, 70, 61, 1.0, "f32"),
                                              (96, 80, 70, 61, 0.37, "f64"), (256, 256, 256, 256, 1.0, "f32"),
                                              (5, 7, 5, 7, 1.0, "f32"), (1, 1, 1, 1, 1.0, "f32")]:
                tdt = torch.float32 if dt == "f32" else torch.float64
                sc = T(3.0 * synth.hash_normal(1, N * 1000 + M, (1, N, M))).to(tdt)
                sm = torch.arange(N)[None] < nv
                tm = torch.arange(M)[None] < mv
                sc = sc.masked_fill(~(sm[:, :, None] & tm[:, None, :]), float("-inf"))
                Z = log_optimal_transport(sc, torch.tensor(alpha, dtype=torch.float32), 3, sm, tm)
                assert Z.dtype == tdt
                f1["logZ_%d_%d_%d_%d_%s_%s" % (N, M, nv, mv, str(alpha).replace(".", "p"), dt)] = Z.numpy()
                # (the mask=None branch of the reference raises: `(ms + ns).log()` on python ints,
                #  3D/models/matching.py:65-67,79 -- quirk Q21; masks are always passed on the path)
            save("sinkhorn", **f1)

        # ---------------- F2: position code ----------------------------------------------------
        cfg_t = ref_config(variant, 20, 200).coarse_transformer
        pe_mod = VolumetricPositionEncoding(cfg_t)
        pair = synth.make_pair(64, 48, C, seed=3)
        ps, pt = T(pair["s_pcd"])[None], T(pair["t_pcd"])[None]
        code = pe_mod(ps)
        feat = T(pair["src_feats"])[None]
        rot = VolumetricPositionEncoding.embed_rotary(feat, code[..., 0], code[..., 1])
        save("pe", cos=code[0, :16, :, 0].numpy(), sin=code[0, :16, :, 1].numpy(), rot=rot[0, :16].numpy())

        # ---------------- F3/F4/F5: attention layer, denoiser, matching head -------------------
        Wnp = Wnp_main
        cfg_t2 = to_attr(dict(cfg_t))
        cfg_t2["layer_types"] = list(synth.LAYER_TYPES)
        den = RepositioningTransformer(cfg_t2)
        den.load_state_dict({k[len("denoising_transformer."):]: T(a) for k, a in Wnp.items()
                             if k.startswith("denoising_transformer.")})
        head = Matching(ref_config(variant, 20, 200).coarse_matching)
        head.load_state_dict({k[len("denoising_coarse_matching."):]: T(a) for k, a in Wnp.items()
                              if k.startswith("denoising_coarse_matching.")})
        den.eval(); head.eval()
        fs, ft = T(pair["src_feats"])[None], T(pair["tgt_feats"])[None]
        sm_full = torch.ones(1, 64, dtype=torch.bool); tm_full = torch.ones(1, 48, dtype=torch.bool)
        sm_part = torch.arange(64)[None] < 50; tm_part = torch.arange(48)[None] < 41
        with torch.no_grad():
            pes, pet = pe_mod(ps), pe_mod(pt)
            lay = den.layers[1]
            f3 = dict(
                self_full=lay(fs, fs, pes, pes, sm_full, sm_full)[0].numpy(),
                cross_full=lay(fs, ft, pes, pet, sm_full, tm_full)[0].numpy(),
                self_mask=lay(fs, fs, pes, pes, sm_part, sm_part)[0].numpy(),
                cross_mask=lay(fs, ft, pes, pet, sm_part, tm_part)[0].numpy())
            save("attn_layer", **f3)
            d = {}
            os_, ot_, pe_s, pe_t = den(fs, ft, ps, pt, sm_full, tm_full, d)
            conf, _ = head(os_, ot_, pe_s, pe_t, sm_full, tm_full, d, pe_type="rotary")
            d = {}
            om_s, om_t, pe_s2, pe_t2 = den(fs, ft, ps, pt, sm_part, tm_part, d)
            confm, _ = head(om_s, om_t, pe_s2, pe_t2, sm_part, tm_part, d, pe_type="rotary")
            save("denoiser", f_s=os_[0].numpy(), f_t=ot_[0].numpy(), conf=conf[0].numpy(),
                 f_s_mask=om_s[0].numpy(), f_t_mask=om_t[0].numpy(), conf_mask=confm[0].numpy())

        # ---------------- F6: SoftProcrustesLayer ----------------------------------------------
        pr = synth.make_pair(128, 128, C, seed=5)
        gtm = np.zeros((128, 128)); gtm[pr["gt_matches"][:, 0], pr["gt_matches"][:, 1]] = 6.0
        sc = T(gtm + synth.hash_normal(5, 77, (128, 128)))[None].float()
        smk = torch.arange(128)[None] < 120; tmk = torch.arange(128)[None] < 111
        Z = log_optimal_transport(sc.masked_fill(~(smk[:, :, None] & tmk[:, None, :]), float("-inf")),
                                  torch.tensor(1.0), 3, smk, tmk)
        cf = Z.exp()[:, :-1, :-1].contiguous()
        f6 = {}
        for mc in (0, 40, 200):
            layer = SoftProcrustesLayer(to_attr(dict(sample_rate=1.0, max_condition_num=mc)))
            R, t, Rf, tf, cond, ok = layer(cf.clone(), T(pr["s_pcd"])[None], T(pr["t_pcd"])[None], smk, tmk)
            f6.update({"R_%d" % mc: R.numpy(), "t_%d" % mc: t.numpy(), "Rf_%d" % mc: Rf.numpy(),
                       "tf_%d" % mc: tf.numpy(), "cond_%d" % mc: cond.numpy(), "ok_%d" % mc: ok.numpy()})
        f6["R_gt"], f6["t_gt"] = pr["R_gt"], pr["t_gt"]
        save("procrustes", **f6)


    Wnp_main = synth.make_weights(C, seed=7, head_gain=HEAD_GAIN)
    if family == "main":
        components()
    # the "soft" family (round 4): the same scenes with the matching head at a checkpoint-like scale (HEAD_GAIN_SOFT: matching logits
    # O(10) instead of thousands), where the reference's own float32 run is within 2e-5 of the float64 evaluation on EVERY entry, so
    # that the loop is held to the reference by a plain 1e-4 bound with an empty exemption list (tests/golden/loop_exemptions.json)
    Wnp = Wnp_main if family == "main" else synth.make_weights(C, seed=7, head_gain=HEAD_GAIN_SOFT)

    # ---------------- F7: the loop through Pipeline.forward --------------------------------
    from models.pipeline import Pipeline

    class StubBackbone(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.feats = None

        def forward(self, data, phase="coarse"):
            return self.feats

    randn_log = []
    injected = {}
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def fake_randn(*a, **k):
        return injected["x_T"].clone()

    def fake_randn_like(x, *a, **k):
        n = injected["noise"][len(randn_log)].to(x.dtype)
        randn_log.append(1)
        return n.clone()

    def run_loop(N, M, nv, mv, steps, max_cond, seed, tag, keep_conf=True):
        cfg = ref_config(variant, steps, max_cond)
        from configs.models import architectures          # the tree's own list (3D/main.py:79)
        cfg.kpfcn_config["architecture"] = architectures[variant]
        torch.randn, torch.randn_like = real_randn, real_randn_like
        model = Pipeline(cfg)
        model.backbone = StubBackbone()
        sd = model.state_dict()
        for k, a in Wnp.items():
            assert k in sd, k
            sd[k] = T(a)
        model.load_state_dict(sd)
        model.eval()
        pr = synth.make_pair(N, M, C, seed=seed)
        feats = torch.cat([T(pr["src_feats"]), T(pr["tgt_feats"])], 0)
        pts = torch.cat([T(pr["s_pcd"]), T(pr["t_pcd"])], 0)
        model.backbone.feats = feats
        data = {"points": [None, None, pts, None], "src_mask": torch.arange(N)[None] < nv,
                "tgt_mask": torch.arange(M)[None] < mv,
                "src_ind_coarse_split": torch.arange(N), "tgt_ind_coarse_split": torch.arange(M),
                "src_ind_coarse": torch.arange(N), "tgt_ind_coarse": torch.arange(N, N + M)}
        injected["x_T"] = T(pr["x_T"])[None]
        injected["noise"] = T(synth.step_noise(N, M, seed, steps))[:, None]
        del randn_log[:]
        x0_log, warp_log = [], []
        orig_head = model.denoising_coarse_matching.forward
        orig_proc = model.denoising_soft_procrustes.forward

        def head_spy(*a, **k):
            r = orig_head(*a, **k)
            x0_log.append(r[0].detach().clone())
            return r

        def proc_spy(*a, **k):
            r = orig_proc(*a, **k)
            warp_log.append([z.detach().clone() for z in r])
            return r

        model.denoising_coarse_matching.forward = head_spy
        model.denoising_soft_procrustes.forward = proc_spy
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        try:
            with torch.no_grad():
                out = model(data)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        conf = out["conf_matrix_pred"]
        rec = dict(conf_dtype=str(conf.dtype), R_s2t_pred=out["R_s2t_pred"].numpy(),
                   t_s2t_pred=out["t_s2t_pred"].numpy(),
                   x0_sum=np.array([float(z.double().sum()) for z in x0_log]),
                   x0_corner=torch.stack([z[0, :16, :16] for z in x0_log]).numpy(),
                   x0_last=x0_log[-1][0].numpy(),
                   R_forwd=torch.stack([w[2][0] for w in warp_log]).numpy(),
                   t_forwd=torch.stack([w[3][0] for w in warp_log]).numpy(),
                   cond=torch.stack([w[4][0] for w in warp_log]).numpy())
        if keep_conf:
            rec["conf"] = conf[0].numpy()
        else:
            rec["conf_sum"] = np.array(float(conf.double().sum()))
            rec["conf_corner"] = conf[0, :32, :32].numpy()
        if "match_pred" in out:
            rec["match_pred"] = out["match_pred"].numpy()
        save("loop_" + tag, **rec)
        c = conf[0].double()
        print(tag, "conf", conf.dtype, "max %.4f" % float(c.max()), "rowmax mean %.4f" % float(c.max(1)[0].mean()),
              "cond", rec["cond"][:3], "R_s2t", rec["R_s2t_pred"].ravel()[:3])

    if family == "soft":
        if variant == "3dmatch":
            run_loop(128, 128, 128, 128, 1, 200, seed=11, tag="soft_n128_s1_mc200")      # cfg1
            run_loop(256, 256, 256, 256, 20, 200, seed=13, tag="soft_n256_s20_mc200")    # cfg2
        else:
            run_loop(512, 512, 470, 391, 20, 40, seed=62, tag="soft_n512_s20_mc40_masked")   # cfg3's pair 0
        return
    if variant == "3dmatch":
        run_loop(128, 128, 128, 128, 1, 200, seed=11, tag="n128_s1_mc200")      # cfg1
        run_loop(128, 128, 128, 128, 20, 0, seed=11, tag="n128_s20_mc0")        # shipped test yaml (Q4)
        run_loop(96, 80, 96, 80, 5, 200, seed=12, tag="n96x80_s5_mc200")        # ragged
        run_loop(256, 256, 256, 256, 20, 200, seed=13, tag="n256_s20_mc200")    # cfg2
    else:
        run_loop(128, 128, 112, 100, 5, 40, seed=21, tag="n128_s5_mc40_masked")
        run_loop(64, 96, 64, 96, 20, 40, seed=22, tag="n64x96_s20_mc40")
        # BASELINE configs[2] at its stated size: one pair of the 8, 512 x 512 (padded from 470 x 391), 20 steps, masks, sigma*xi
        run_loop(512, 512, 470, 391, 20, 40, seed=62, tag="n512_s20_mc40_masked")


def run_2d3d():
    """row a10: the reference's CrossModalFusionModule / Matching / SoftProcrustesLayer / log_optimal_transport /
    mutual_topk_select composed in the order of MATR2D3D.forward's reverse sampling (EXP/model.py:637-694, 830-846);
    MATR2D3D itself cannot be built offline (DINOv2 / Depth-Anything downloads)."""
    import torch
    for m in ("vision3d.ext", "ipdb", "open3d", "cv2", "easydict", "pykeops", "pykeops.torch", "pytorch3d", "pytorch3d.ops"):
        sys.modules[m] = MagicMock()
    torch.Tensor.cuda = lambda self, *a, **k: self
    tree = os.path.join(REF, "Diff-Reg-2d3d")
    exp = os.path.join(tree, "experiments", "2d3dmatr.rgbdv2.stage4.level3.stage1")
    sys.path.insert(0, tree)
    sys.path.insert(0, exp)
    os.chdir(exp)
    torch.set_num_threads(8)
    from diffreg_hip import synth
    from fusion_module import CrossModalFusionModule
    from matching import Matching, log_optimal_transport
    from procrustes import SoftProcrustesLayer
    from vision3d.ops import mutual_topk_select
    sys.path.insert(0, ROOT)
    from oracle.diffreg_oracle import diffusion_schedule, time_pairs

    v = synth.VARIANTS["2d3d"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    scout = os.environ.get("SCOUT_2D3D")        # "gain,xT scale,seed,mc[,N,M]": print the K-th boundary gaps of one run, write nothing
    sc_gain, sc_xt = (float(scout.split(",")[0]), float(scout.split(",")[1])) if scout else (HEAD_GAIN_2D3D, 1.0)
    Wnp = synth.make_weights_2d3d(seed=9, head_gain=sc_gain)
    fus = CrossModalFusionModule(v["img_dim"], v["pcd_dim"], v["C"], v["C"], v["H"], ["self", "cross"] * 3, use_embedding=True)
    fus.load_state_dict({k[len("denoising_transformer."):]: T(a) for k, a in Wnp.items() if k.startswith("denoising_transformer.")})
    mcfg = to_attr(dict(feature_dim=v["C"], confidence_threshold=0.2, entangled=False, dsmax_temperature=0.1,
                        match_type="sinkhorn", skh_init_bin_score=1.0, skh_iters=3, skh_prefilter=False))
    head = Matching(mcfg)
    head.load_state_dict({k[len("denoising_coarse_matching."):]: T(a) for k, a in Wnp.items() if k.startswith("denoising_coarse_matching.")})
    fus.eval(); head.eval()
    ac, sra, srm1 = diffusion_schedule()
    save = lambda name, **kw: np.savez_compressed(os.path.join(OUT, "2d3d_%s.npz" % name), **kw)

    def run(N, M, nv, mv, mv_da, steps, mc, seed, tag, compact=False, xt_scale=None):
        proc = SoftProcrustesLayer(to_attr(dict(sample_rate=1.0, max_condition_num=mc)))
        pr = synth.make_pair_2d3d(N, M, seed, weights=Wnp)
        g = lambda k: T(pr[k])[None]
        src_mask = torch.arange(N)[None] < nv
        tgt_mask = torch.arange(M)[None] < mv
        tgt_mask_da = torch.arange(M)[None] < mv_da
        xts = sc_xt if xt_scale is None else xt_scale
        x = g("x_T").clone() * xts
        rec = dict(x0=[], Rf=[], tf=[], cond=[], gap=[])
        Ksel = int(max(nv, mv_da) * 1.0)                # top-K of SoftProcrustesLayer (EXP/procrustes.py:61-62, sample_rate 1)
        with torch.no_grad():
            f_img0, f_pcd0 = fus(g("img_feats"), g("img_dino"), g("img_pixels"), g("pcd_feats"), g("s_pcd"))
            c0, _, _, _ = head(f_pcd0, f_img0, src_mask, tgt_mask, True)
            for (t, tn) in time_pairs(steps):
                x.masked_fill_(~(src_mask[..., None] * tgt_mask_da[:, None]).bool(), float("-inf"))
                Z = log_optimal_transport(x, head.bin_score, head.skh_iters, src_mask, tgt_mask_da)
                cd = Z.exp()[:, :-1, :-1].contiguous().type(torch.float32)
                R, tt, Rf, tf, cond, ok = proc(cd, g("s_pcd"), g("t_pcd_da"), src_mask, tgt_mask_da)
                vk = cd.flatten().topk(Ksel + 1)[0]
                rec["gap"].append(float((vk[Ksel - 1] - vk[Ksel]) / vk[Ksel - 1]))   # 0 = exact tie at the K-th boundary: torch.topk's pick is implementation-defined
                warped = (torch.matmul(Rf.type(torch.float32), g("s_pcd").transpose(1, 2)) + tf.type(torch.float32)).transpose(1, 2)
                f_img, f_pcd = fus(g("img_feats"), g("img_dino"), g("img_pixels"), g("pcd_feats"), warped)
                x0, _, _, _ = head(f_pcd, f_img, src_mask, tgt_mask, True)
                tc = torch.full((1,), t, dtype=torch.long)
                pred = (sra.gather(-1, tc).reshape(1, 1, 1) * x - x0) / srm1.gather(-1, tc).reshape(1, 1, 1)
                a, an = ac[t], ac[tn]
                sigma = 1.0 * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                c = (1 - an - sigma ** 2).sqrt()
                x = x0 * an.sqrt() + c * pred
                rec["x0"].append(x0[0].clone()); rec["Rf"].append(Rf[0].clone()); rec["tf"].append(tf[0].clone()); rec["cond"].append(cond[0].clone())
            sim = x
            sim.masked_fill_(~(src_mask[..., None] * tgt_mask[:, None]).bool(), float("-inf"))
            Z = log_optimal_transport(sim, head.bin_score, head.skh_iters, src_mask, tgt_mask)
            conf = Z.exp()[:, :-1, :-1].contiguous()
            i, j, sc = mutual_topk_select(conf.squeeze(0), 1, largest=True, threshold=None, mutual=False)
        x0s = torch.stack(rec["x0"])
        if scout:
            print("SCOUT", scout, "K-th boundary gaps", ["%.2e" % g_ for g_ in rec["gap"]], "cond", ["%.1f" % float(c) for c in rec["cond"]], flush=True)
            return
        if compact:
            # BASELINE configs[4] size: the matrices are 8 / 16 MB -- the fixture keeps every 8th row and column, the row / column sums of
            # the whole matrices (any wrong entry above the tolerance moves one of each), and the per-step poses
            c64 = conf[0].double()
            fin = torch.isfinite(c64)
            save("loop_" + tag, conf_sub=c64[::8, ::8].numpy(), conf_rowsum=c64.sum(1).numpy(), conf_colsum=c64.sum(0).numpy(),
                 conf_dtype=str(conf.dtype), conf_all_finite=bool(fin.all()), x0_corner=x0s[:, :16, :16].numpy(),
                 x0_last_sub=x0s[-1][::8, ::8].numpy(), x0_last_rowsum=x0s[-1].double().sum(1).numpy(), x0_last_colsum=x0s[-1].double().sum(0).numpy(),
                 x0_sum=x0s.double().sum((1, 2)).numpy(), R_forwd=torch.stack(rec["Rf"]).numpy(), t_forwd=torch.stack(rec["tf"]).numpy(),
                 cond=torch.stack(rec["cond"]).numpy(), kth_gap_rel=np.asarray(rec["gap"]), match_i=i.numpy(), match_j=j.numpy(), xt_scale=np.float32(xts))
            print(tag, "conf", conf.dtype, "cond", [float(c) for c in rec["cond"]], "K-th boundary gaps", rec["gap"])
            return
        save("loop_" + tag, f_img0=f_img0[0].numpy(), f_pcd0=f_pcd0[0].numpy(), conf0=c0[0].numpy(), conf=conf[0].numpy(),
             conf_dtype=str(conf.dtype), x0_corner=x0s[:, :16, :16].numpy(), x0_last=x0s[-1].numpy(),
             x0_sum=x0s.double().sum((1, 2)).numpy(), R_forwd=torch.stack(rec["Rf"]).numpy(), t_forwd=torch.stack(rec["tf"]).numpy(),
             cond=torch.stack(rec["cond"]).numpy(), match_i=i.numpy(), match_j=j.numpy())
        print(tag, "conf", conf.dtype, "x0 rowmax mean %.3f" % float(x0s[-1].max(1)[0].mean()), "cond", rec["cond"][:3])

    if scout:
        a = scout.split(",")
        N_, M_ = (int(a[4]), int(a[5])) if len(a) > 5 else (1024, 2048)
        run(N_, M_, N_ - 24, M_ - 48, M_ - 148, 10, float(a[3]), int(a[2]), "scout", compact=True)
        return
    if os.environ.get("MINT_CFG5_WARP_ONLY") == "1":
        run(1024, 2048, 1000, 2000, 1900, 10, 200, 51, "n1024x2048_s10_mc200_xt03_masked", compact=True, xt_scale=0.3)
        return
    run(96, 160, 90, 150, 141, 3, 200, 31, "n96x160_s3_masked")
    # (the random-weight fusion module gives fairly flat matrices, whose top-K is unstable over many steps with the
    #  warp active; the long run therefore uses the identity warp, the short masked run exercises the Procrustes feedback)
    run(128, 192, 128, 192, 192, 10, 0, 32, "n128x192_s10_mc0")
    # BASELINE configs[4] at its stated size (1024 point nodes x 2048 patches, 10 steps, padding masks on both sides and a different
    # tgt_mask_da): pins the large-tile Sinkhorn / top-K kernels to the reference's own components.  Identity warp (max_condition_num 0),
    # like the long small run: with these synthetic weights the noised matrix of step 1 is so flat that its K-th and (K+1)-th confidences
    # are EQUAL in float32 for every seed tried (kth_gap_rel in the fixture; 41 of 41 seeds): with the warp fed back, torch.topk's
    # implementation-defined pick among equal values would decide the rest of the trajectory.  cond of a tied step is held loosely by the tests.
    if os.environ.get("MINT_CFG5", "1") == "1":
        run(1024, 2048, 1000, 2000, 1900, 10, 0, 51, "n1024x2048_s10_mc0_masked", compact=True)
    # ... and WITH the warp fed back (max_condition_num 200; x_T scaled by 0.3 so that step 1's matrix is not clipped flat: every step's K-th
    # boundary gap is non-zero, 1e-7 .. 2e-5 relative).  At K = 2 000 of 2 M entries the gap is one ulp whatever the scene (the spacing of order
    # statistics), so WHICH of two equal-confidence correspondences is taken is still implementation noise -- but either choice moves the weighted
    # Kabsch fit far below 1e-4: the fixture pins the poses of all 10 steps, the state and the read-out of the warp-active loop to the REFERENCE.
    if os.environ.get("MINT_CFG5_WARP", "1") == "1":
        run(1024, 2048, 1000, 2000, 1900, 10, 200, 51, "n1024x2048_s10_mc200_xt03_masked", compact=True, xt_scale=0.3)


HEAD_GAIN_2D3D = 16.0
HEAD_GAIN = 24.0   # sharpens x_start so the synthetic scenes give near-permutation matrices (section 8c F7)
HEAD_GAIN_SOFT = 3.0   # the "soft" family: logits O(10), nothing ill-conditioned (no exemptions), still row maxima ~ 0.57

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_2d3d() if sys.argv[1] == "2d3d" else run_tree(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "main")
    else:
        for v in list(TREES) + ["2d3d"]:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), v])
        for v in TREES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), v, "soft"])
