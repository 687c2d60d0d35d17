"""Mint golden vectors for the forward half of the training branch by RUNNING THE REFERENCE
(build container only; needs /root/reference).

    python oracle/make_golden_train.py          # writes tests/golden/train_forward.npz

What runs: the reference's own Pipeline.forward with `model.train()` (3D/models/pipeline.py:182-216: coarse transformer with its
positioning layer, coarse matching, soft Procrustes, GT-matrix noising through q_sample, denoising transformer + matching on
the warped source) and the reference's MatchMotionLoss.forward on the dict it returns (3D/models/loss.py:73-170), plus
MatchMotionLoss.compute_correspondence_loss / compute_match_recall alone on corner cases.  Shims: the ones of
oracle/make_golden.py (open3d / easydict / tensorboardX / cv2 / nibabel mocks, attribute dict, stub backbone); torch.randint and
torch.randn are replaced by injected values for the two draws of the branch (the time step, the noise matrix).
Inputs and weights come from diffreg_hip.synth (integer hash): only reference OUTPUTS are stored.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "train_forward.npz")
TREE = "/root/reference/Diff-Reg-3dmatch"
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)

# (tag, B, N, M, seed of pair 0, time step, max_condition_num)
# seeds: chosen so that none of the three Procrustes fits of a case has equal K-th and (K+1)-th confidences (asserted below).  Exact ties
# are common in these matrices (a target column nobody matches gets the same Sinkhorn mass from every row that sits in its dustbin), and
# torch.topk leaves the choice among equal values to the implementation -- the reference's own CPU and CUDA runs differ there.
CASES = [("b1", 1, 96, 80, 50, 137, 200.0), ("b2", 2, 64, 64, 60, 640, 200.0)]
LOSS_CFG = dict(focal_alpha=0.25, focal_gamma=2.0, pos_weight=1.0, neg_weight=1.0, motion_loss_type="L1", motion_weight=0.0, match_weight=1,
                match_type="sinkhorn", positioning_type="procrustes", confidence_threshold_metric=0.05, mutual_nearest=False,
                inlier_thr=0.1, fmr_thr=0.05, registration_threshold=0.2, dataset="3dmatch")


def main():
    import torch
    from oracle.make_golden import ref_config, to_attr, HEAD_GAIN
    sys.modules["open3d"] = MagicMock()
    for m in ("easydict", "tensorboardX", "nibabel", "nibabel.quaternions", "cv2"):
        sys.modules.setdefault(m, MagicMock())
    torch.Tensor.cuda = lambda self, *a, **k: self
    os.chdir(TREE)
    sys.path.insert(0, TREE)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from diffreg_hip import synth
    from models.pipeline import Pipeline
    from models.loss import MatchMotionLoss
    from configs.models import architectures
    from models.procrustes import SoftProcrustesLayer
    real_proc = SoftProcrustesLayer.forward

    def tie_checked(self, conf_matrix, src_pcd, tgt_pcd, src_mask, tgt_mask):
        K = max(conf_matrix.shape[1:])
        for b in range(conf_matrix.shape[0]):
            w = torch.topk(conf_matrix[b].reshape(-1), K + 1)[0]
            assert w[K - 1] != w[K], "tie at the K-th confidence: pick another seed"
        return real_proc(self, conf_matrix, src_pcd, tgt_pcd, src_mask, tgt_mask)
    SoftProcrustesLayer.forward = tie_checked

    v = synth.VARIANTS["3dmatch"]
    C = v["C"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    Wnp = dict(synth.make_weights(C, seed=7, head_gain=HEAD_GAIN))
    Wnp.update(synth.make_weights_coarse(C, seed=17, head_gain=HEAD_GAIN))

    class StubBackbone(torch.nn.Module):
        feats = None

        def forward(self, data, phase="coarse"):
            return self.feats

    real_randn, real_randint = torch.randn, torch.randint
    out = {}
    for tag, B, N, M, seed, ts, mc in CASES:
        cfg = ref_config("3dmatch", 20, mc)
        cfg.kpfcn_config["architecture"] = architectures["3dmatch"]
        model = Pipeline(cfg)
        model.backbone = StubBackbone()
        sd = model.state_dict()
        for k, a in Wnp.items():
            assert k in sd, k
            sd[k] = T(a)
        missing = [k for k in sd if k not in Wnp and not k.startswith("backbone") and "alphas_cumprod" not in k]
        assert not missing, missing[:5]
        model.load_state_dict(sd)
        model.train()
        prs = [synth.make_pair(N, M, C, seed=seed + b) for b in range(B)]
        feats = torch.cat([T(p["src_feats"]) for p in prs] + [T(p["tgt_feats"]) for p in prs], 0)
        pts = torch.cat([T(p["s_pcd"]) for p in prs] + [T(p["t_pcd"]) for p in prs], 0)
        model.backbone.feats = feats
        data = {"points": [None, None, pts, None], "src_mask": torch.ones(B, N, dtype=torch.bool), "tgt_mask": torch.ones(B, M, dtype=torch.bool),
                "src_ind_coarse_split": torch.arange(B * N), "tgt_ind_coarse_split": torch.arange(B * M),
                "src_ind_coarse": torch.arange(B * N), "tgt_ind_coarse": torch.arange(B * N, B * (N + M)),
                "coarse_matches": [T(p["gt_matches"]).t().contiguous() for p in prs],
                "batched_rot": torch.stack([T(p["R_gt"]).float() for p in prs]),
                "batched_trn": torch.stack([T(p["t_gt"]).float().view(3, 1) for p in prs])}
        randn = T(synth.hash_normal(seed, 900, (B, N, M)).astype(np.float32))
        randn[0, 0, 0] = 0.0                                   # the 0 / 0 = nan -> 0 entry of nan_to_num
        captured = {}
        orig = model.get_warped_from_noising_matching

        def spy(s_pcd, t_pcd, src_mask, tgt_mask, matrix_gt_disturbed):
            captured["noised"] = matrix_gt_disturbed.detach().clone()
            r = orig(s_pcd, t_pcd, src_mask, tgt_mask, matrix_gt_disturbed)
            captured["src_warped"] = r[0].detach().clone()
            return r
        model.get_warped_from_noising_matching = spy
        torch.randn = lambda *a, **k: randn.clone()
        torch.randint = lambda *a, **k: torch.tensor([ts])
        try:
            with torch.no_grad():
                res = model(data)
        finally:
            torch.randn, torch.randint = real_randn, real_randint
        assert captured["noised"].dtype == torch.float64
        rec = dict(ts=np.int64(ts), noised=captured["noised"].numpy(), src_warped=captured["src_warped"].numpy(),
                   conf_matrix_pred=res["conf_matrix_pred"].numpy(), coarse_match_pred=res["coarse_match_pred"].numpy(),
                   R_s2t_pred=res["R_s2t_pred"].numpy(), t_s2t_pred=res["t_s2t_pred"].numpy(),
                   conf_matrix_gt_hat=res["conf_matrix_gt_hat"].numpy(), coarse_match_gt_hat=res["coarse_match_gt_hat"].numpy())
        for mot_w in (0.0, 1.0):
            crit = MatchMotionLoss(dict(LOSS_CFG, motion_weight=mot_w))
            info = crit(res)
            for k, val in info.items():
                rec["loss_mot%d_%s" % (int(mot_w), k)] = np.asarray(float(val), dtype=np.float64)
        for k, a in rec.items():
            out[tag + "_" + k] = a
        print(tag, "noised", rec["noised"].dtype, "min %.3f max %.3f" % (rec["noised"].min(), rec["noised"].max()), "conf_pred", res["conf_matrix_pred"].dtype,
              {k: float(a) for k, a in rec.items() if k.startswith("loss_")})

    # ---- compute_correspondence_loss / compute_match_recall alone, corner cases included (loss.py:273-345)
    P, N, M = 2, 40, 56
    conf = T(synth.hash_u01(5, 1, P * N * M).reshape(P, N, M).astype(np.float32))
    conf[0, 0, :4] = T(np.array([0.0, 1.0, 1e-7, 1 - 1e-8], dtype=np.float32))          # both clamps
    gt = torch.zeros(P, N, M)
    gi = torch.from_numpy(synth.hash_u01(5, 2, 60)).mul(P * N * M).long()
    gt.view(-1)[gi] = 1.0
    weight = T(synth.hash_u01(5, 3, P * N * M).reshape(P, N, M).astype(np.float32))
    for mt in ("sinkhorn", "dual_softmax"):
        for gamma, alpha, pw, nw in ((2.0, 0.25, 1.0, 1.0), (1.5, 0.4, 0.7, 2.0)):
            crit = MatchMotionLoss(dict(LOSS_CFG, match_type=mt, focal_gamma=gamma, focal_alpha=alpha, pos_weight=pw, neg_weight=nw))
            nm = "focal_%s_g%s" % (mt, str(gamma).replace(".", "p"))
            out[nm] = np.float64(float(crit.compute_correspondence_loss(conf.clone(), gt.clone(), weight=weight.clone())))
            out[nm + "_nopos"] = np.float64(float(crit.compute_correspondence_loss(conf.clone(), torch.zeros_like(gt), weight=weight.clone())))
            out[nm + "_noneg"] = np.float64(float(crit.compute_correspondence_loss(conf.clone(), torch.ones_like(gt), weight=weight.clone())))
    pred = torch.stack([gi[:40] // (N * M), (gi[:40] // M) % N, gi[:40] % M], 1)           # 40 true entries ...
    extra = torch.tensor([[0, 1, 1], [1, 2, 3], [1, 2, 3], [0, 39, 55]])                    # ... and 4 more (one repeated)
    pred = torch.cat([pred, extra], 0)
    r, p = MatchMotionLoss.compute_match_recall(gt, pred)
    out["recall"], out["precision"] = np.float64(float(r)), np.float64(float(p))
    out["recall_pred"] = pred.numpy()
    out["focal_gt_index"] = gi.numpy()
    # ---- backward of the matching head's loss: autograd through the reference's own log_optimal_transport (models/matching.py:61-93),
    #      exp + slice (:213-214) and compute_correspondence_loss (models/loss.py:273-314)
    from models.matching import log_optimal_transport
    from tests.helpers import train_backward_case
    bw = {}
    for tag in ("full", "masked", "big"):
        sc, gt_b, sm, tm = train_backward_case(tag)
        sc = sc.masked_fill(~(sm[:, :, None] & tm[:, None, :]), float("-inf")).requires_grad_(True)
        alpha = torch.tensor(1.0, requires_grad=True)
        with torch.enable_grad():
            Z = log_optimal_transport(sc, alpha, 3, sm, tm)
            conf = Z.exp()[:, :-1, :-1].contiguous()
            conf.retain_grad()
            crit = MatchMotionLoss(dict(LOSS_CFG))
            loss = crit.compute_correspondence_loss(conf, gt_b.clone())
            loss.backward()
        bw[tag + "_loss"] = np.float64(float(loss))
        bw[tag + "_conf"] = conf.detach().numpy()
        bw[tag + "_grad_conf"] = conf.grad.numpy()
        bw[tag + "_grad_scores"] = sc.grad.numpy()
        bw[tag + "_grad_bin_score"] = np.float64(float(alpha.grad))
        print("backward", tag, "loss %.6f" % float(loss), "|grad_scores| max %.3e" % float(sc.grad.abs().max()), "grad_bin_score %.6e" % float(alpha.grad))
    # ---- the whole matching head: autograd through the reference's Matching.forward (models/matching.py:164-216, sinkhorn, rotary code) and the
    #      focal loss, down to the features, the shared projection weight (quirk Q1) and bin_score
    from models.matching import Matching
    from models.position_encoding import VolumetricPositionEncoding
    cfgm = ref_config("3dmatch", 20, 200.0)
    head = Matching(cfgm.coarse_matching)
    head.load_state_dict({k[len("coarse_matching."):]: T(a) for k, a in Wnp.items() if k.startswith("coarse_matching.")})
    pe_mod = VolumetricPositionEncoding(cfgm.coarse_transformer)
    c1 = None
    from tests.helpers import train_case
    c1 = train_case("b2")
    fs = (c1["f_s"] * 0.2).clone().requires_grad_(True)
    ft = (c1["f_t"] * 0.2).clone().requires_grad_(True)
    smk = torch.arange(c1["N"])[None].expand(c1["B"], -1) < 57
    tmk = torch.arange(c1["M"])[None].expand(c1["B"], -1) < 60
    with torch.enable_grad():
        conf_h, _ = head(fs, ft, pe_mod(c1["p_s"]), pe_mod(c1["p_t"]), smk, tmk, {}, pe_type="rotary")
        gt_h = torch.zeros_like(conf_h)
        for b_, m_ in enumerate(c1["matches"]):
            keep = (m_[0] < 57) & (m_[1] < 60)
            gt_h[b_][m_[0][keep], m_[1][keep]] = 1
        loss_h = MatchMotionLoss(dict(LOSS_CFG)).compute_correspondence_loss(conf_h, gt_h)
        loss_h.backward()
    bw.update(head_loss=np.float64(float(loss_h)), head_conf=conf_h.detach().numpy(), head_gt=gt_h.numpy(), head_grad_src=fs.grad.numpy(),
              head_grad_tgt=ft.grad.numpy(), head_grad_weight=head.src_proj.weight.grad.numpy(), head_grad_bin_score=np.float64(float(head.bin_score.grad)))
    assert head.tgt_proj.weight.grad is None                     # (allocated, never used: quirk Q1)
    print("backward head loss %.6f |g_src| %.3e |g_W| %.3e g_bin %.3e" % (float(loss_h), float(fs.grad.abs().max()), float(head.src_proj.weight.grad.abs().max()),
                                                                         float(head.bin_score.grad)))
    # ---- one GeometryAttentionLayer (models/transformero.py:43-96): autograd through the reference's module, cross attention with masks
    from models.transformero import GeometryAttentionLayer
    lay = GeometryAttentionLayer(cfgm.coarse_transformer)
    pre = "denoising_transformer.layers.1."
    lay.load_state_dict({k[len(pre):]: T(a) for k, a in Wnp.items() if k.startswith(pre)})
    pr = synth.make_pair(64, 48, C, seed=3)
    xs = (T(pr["src_feats"])[None] * 0.5).clone().requires_grad_(True)
    ys = (T(pr["tgt_feats"])[None] * 0.5).clone().requires_grad_(True)
    pxs, pys = pe_mod(T(pr["s_pcd"])[None]), pe_mod(T(pr["t_pcd"])[None])
    xm, ym = torch.arange(64)[None] < 50, torch.arange(48)[None] < 41
    Rw = T(synth.hash_normal(3, 800, (1, 64, C)).astype(np.float32))
    with torch.enable_grad():
        e = lay(xs, ys, pxs, pys, xm, ym)
        (e * Rw).sum().backward()
    bw.update(layer_out=e.detach().numpy(), layer_grad_x=xs.grad.numpy(), layer_grad_source=ys.grad.numpy())
    for k, prm in lay.named_parameters():
        gq = prm.grad
        bw["layer_grad_" + k] = (gq[::6, ::6] if gq.dim() == 2 else gq).numpy()
        bw["layer_gradnorm_" + k] = np.float64(float(gq.double().norm()))
    print("backward layer |e| %.3f |g_x| %.3e |g_src| %.3e" % (float(e.abs().max()), float(xs.grad.abs().max()), float(ys.grad.abs().max())))
    # ---- the denoising branch of the training loss end to end (pipeline.py:209-212 + loss.py:160-163): reference modules, reference autograd
    cfg_d = ref_config("3dmatch", 20, 200.0)
    cfg_d.kpfcn_config["architecture"] = architectures["3dmatch"]
    pipe = Pipeline(cfg_d)
    sd = pipe.state_dict()
    for k, a in Wnp.items():
        sd[k] = T(a)
    pipe.load_state_dict(sd)
    cb = train_case("b1")
    fs_d = (cb["f_s"] * 0.5).clone().requires_grad_(True)
    ft_d = (cb["f_t"] * 0.5).clone().requires_grad_(True)
    warped = T(out["b1_src_warped"])
    with torch.enable_grad():
        s_n, t_n, pe_s, pe_t = pipe.denoising_transformer(fs_d, ft_d, warped, cb["p_t"], cb["src_mask"], cb["tgt_mask"], {})
        hat, _ = pipe.denoising_coarse_matching(s_n, t_n, pe_s, pe_t, cb["src_mask"], cb["tgt_mask"], {}, pe_type="rotary")
        gt_d = torch.zeros_like(hat)
        gt_d[0][cb["matches"][0][0], cb["matches"][0][1]] = 1
        loss_d = MatchMotionLoss(dict(LOSS_CFG)).compute_correspondence_loss(hat, gt_d)
        loss_d.backward()
    bw.update(branch_loss=np.float64(float(loss_d)), branch_conf=hat.detach().numpy(), branch_grad_src=fs_d.grad.numpy(), branch_grad_tgt=ft_d.grad.numpy())
    for k, prm in list(pipe.denoising_transformer.named_parameters()) + [("head." + k2, p2) for k2, p2 in pipe.denoising_coarse_matching.named_parameters()]:
        if prm.grad is not None:
            bw["branch_gradnorm_" + k] = np.float64(float(prm.grad.double().norm()))
    bw["branch_grad_layers.0.q_proj.weight"] = pipe.denoising_transformer.layers[0].q_proj.weight.grad[::6, ::6].numpy()
    bw["branch_grad_layers.5.mlp.2.weight"] = pipe.denoising_transformer.layers[5].mlp[2].weight.grad[::6, ::6].numpy()
    print("backward branch loss %.6f |g_src| %.3e" % (float(loss_d), float(fs_d.grad.abs().max())), "params with grad:", sum(1 for k in bw if k.startswith("branch_gradnorm_")))
    # ---- the non-denoising branch with the motion term (pipeline.py:184-196 + loss.py:97-128, motion_weight 0.1 as 4DMatch trains): coarse_transformer
    #      (its position codes are detached by the reference, position_encoding.py:83-84) + coarse_matching + soft_procrustes (host SVD, differentiable)
    fs_c = (cb["f_s"] * 0.5).clone().requires_grad_(True)
    ft_c = (cb["f_t"] * 0.5).clone().requires_grad_(True)
    for prm in pipe.parameters():
        prm.grad = None
    ov = torch.zeros(1, cb["N"], dtype=torch.bool)
    ov[0][cb["matches"][0][0]] = True
    with torch.enable_grad():
        a_s, a_t, pe_s2, pe_t2 = pipe.coarse_transformer(fs_c, ft_c, cb["p_s"], cb["p_t"], cb["src_mask"], cb["tgt_mask"], {})
        conf_c, _ = pipe.coarse_matching(a_s, a_t, pe_s2, pe_t2, cb["src_mask"], cb["tgt_mask"], {}, pe_type="rotary")
        R_c, t_c, _, _, _, _ = pipe.soft_procrustes(conf_c, cb["p_s"], cb["p_t"], cb["src_mask"], cb["tgt_mask"])
        focal_c = MatchMotionLoss(dict(LOSS_CFG)).compute_correspondence_loss(conf_c, gt_d)
        wp = (torch.matmul(R_c, cb["p_s"].transpose(1, 2)) + t_c).transpose(1, 2)                       # loss.py:113-127
        wg = (torch.matmul(cb["R_gt"], cb["p_s"].transpose(1, 2)) + cb["t_gt"]).transpose(1, 2)
        l1_c = torch.sum(torch.abs((wp - cb["p_s"]) - (wg - cb["p_s"])), 2)[ov].mean()
        loss_c = focal_c + 0.1 * l1_c
        loss_c.backward()
    bw.update(coarse_loss=np.float64(float(loss_c)), coarse_focal=np.float64(float(focal_c)), coarse_l1=np.float64(float(l1_c)), coarse_conf=conf_c.detach().numpy(),
              coarse_R=R_c.detach().numpy(), coarse_t=t_c.detach().numpy(), coarse_grad_src=fs_c.grad.numpy(), coarse_grad_tgt=ft_c.grad.numpy())
    n_c = 0
    for k, prm in list(pipe.coarse_transformer.named_parameters()) + [("head." + k2, p2) for k2, p2 in pipe.coarse_matching.named_parameters()]:
        if prm.grad is not None:
            bw["coarse_gradnorm_" + k] = np.float64(float(prm.grad.double().norm()))
            n_c += 1
    assert pipe.coarse_transformer.layers[2][0].src_proj.weight.grad is None        # the positioning layer's Matching gets no gradient (detached code)
    print("backward coarse loss %.6f (focal %.6f, l1 %.6f) |g_src| %.3e params with grad: %d" % (float(loss_c), float(focal_c), float(l1_c), float(fs_c.grad.abs().max()), n_c))
    np.savez_compressed(OUT.replace("train_forward", "train_backward"), **bw)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
