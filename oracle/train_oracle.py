"""ORACLE -- CPU restatement of the forward half of Diff-Reg's training branch.  TEST INFRASTRUCTURE.

Only tests/ may import this module; the product path (diff-reg_amd/) never does.  Torch CPU tensors in the dtypes the
reference ends up using.  Citations are into /root/reference/Diff-Reg-3dmatch (3D/).

PINNING: against outputs of the reference itself (its Pipeline.forward under model.train() and its MatchMotionLoss), minted in
the build container by oracle/make_golden_train.py and committed as tests/golden/train_forward.npz;
tests/test_train_oracle.py checks every function below against them.
"""
import torch

from oracle import diffreg_oracle as orc


def match_matrix(matches, P, N, M):
    """3D/models/loss.py:316-320 (match_2_conf_matrix), pipeline.py:203-206: matches = per pair [2, K] index tensors"""
    gt = torch.zeros(P, N, M)
    for b, m in enumerate(matches):
        gt[b][m[0], m[1]] = 1
    return gt


def gt_noising(matrix_gt, randn, t, T=1000):
    """3D/models/pipeline.py:209-214 with q_sample (:84-95) -> float64 [P,N,M]"""
    noise = (randn.abs() % 1) * (randn.abs() / randn) * 1.5                       # :210 (float32; 0 / 0 = nan at randn == 0)
    ac = orc.diffusion_schedule(T)[0]                                             # float64 (:83-93)
    x = ac[t].sqrt().view(1, 1, 1) * matrix_gt + (1.0 - ac[t]).sqrt().view(1, 1, 1) * noise      # :93-95: float64 by promotion
    x = torch.nan_to_num(x, nan=0)                                                # :213
    return x - x.min()                                                            # :214: ONE minimum for the whole batch


def focal_loss(conf, conf_gt, weight=None, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0, match_type="sinkhorn"):
    """3D/models/loss.py:273-314"""
    pos, neg = conf_gt == 1, conf_gt == 0
    if weight is not None:
        weight = weight.clone()
    if not pos.any():                                                             # :287-291
        pos = pos.clone(); pos[0, 0, 0] = True
        if weight is not None:
            weight[0, 0, 0] = 0.
        pos_w = 0.
    if not neg.any():                                                             # :292-296
        neg = neg.clone(); neg[0, 0, 0] = True
        if weight is not None:
            weight[0, 0, 0] = 0.
        neg_w = 0.
    c = torch.clamp(conf, 1e-6, 1 - 1e-6)
    if match_type == "dual_softmax":                                              # :303-309
        lp = -alpha * torch.pow(1 - c[pos], gamma) * c[pos].log()
        if weight is not None:
            lp = lp * weight[pos]
        return pos_w * lp.mean()
    lp = -alpha * torch.pow(1 - c[pos], gamma) * c[pos].log()                     # :311-314
    ln = -alpha * torch.pow(c[neg], gamma) * (1 - c[neg]).log()
    return pos_w * lp.mean() + neg_w * ln.mean()


def match_recall(conf_gt, match_pred):
    """3D/models/loss.py:323-345"""
    pred = torch.zeros_like(conf_gt)
    pred[match_pred[:, 0], match_pred[:, 1], match_pred[:, 2]] = 1.
    tp = ((pred == conf_gt) * conf_gt).sum()
    return tp / conf_gt.sum(), tp / max(len(match_pred), 1)


def motion_l1(s_pcd, R_pred, t_pred, R_gt, t_gt, overlap_mask, flow=None):
    """3D/models/loss.py:108-128"""
    wp = (R_pred @ s_pcd.transpose(1, 2) + t_pred).transpose(1, 2)
    src = s_pcd + flow if flow is not None else s_pcd
    wg = (R_gt @ src.transpose(1, 2) + t_gt).transpose(1, 2)
    e1 = ((wp - s_pcd) - (wg - s_pcd)).abs().sum(2)
    return e1[overlap_mask].mean()


def coarse_transformer(W, cfg, f_s, f_t, p_s, p_t, mask_s, mask_t, max_cond, prefix="coarse_transformer."):
    """RepositioningTransformer.forward with layer_types [self, cross, positioning, self, cross] (3D/models/transformero.py:143-205):
    the positioning layer = Matching + SoftProcrustesLayer of layers.2, then the position code of the re-posed source"""
    C, H = cfg["C"], cfg["H"]
    pe_s = orc.vol_pe(p_s, C, cfg["origin"], cfg["voxel"])
    pe_t = orc.vol_pe(p_t, C, cfg["origin"], cfg["voxel"])
    for l, name in enumerate(["self", "cross", "positioning", "self", "cross"]):
        pre = prefix + "layers.%d." % l
        if name == "self":
            f_s = orc.attention_layer(W, pre, f_s, f_s, pe_s, pe_s, mask_s, mask_s, H)
            f_t = orc.attention_layer(W, pre, f_t, f_t, pe_t, pe_t, mask_t, mask_t, H)
        elif name == "cross":
            f_s = orc.attention_layer(W, pre, f_s, f_t, pe_s, pe_t, mask_s, mask_t, H)
            f_t = orc.attention_layer(W, pre, f_t, f_s, pe_t, pe_s, mask_t, mask_s, H)
        else:
            conf = orc.match_head(W, cfg, f_s, f_t, pe_s, pe_t, mask_s, mask_t, prefix=pre + "0.")
            R, t, Rf, tf, cond, ok = orc.procrustes(conf.float(), p_s, p_t, mask_s, mask_t, cfg["sample_rate"], max_cond)
            warped = (Rf.float() @ p_s.transpose(1, 2) + tf.float()).transpose(1, 2)
            pe_s = orc.vol_pe(warped, C, cfg["origin"], cfg["voxel"])
    return f_s, f_t, pe_s, pe_t


def training_forward(W, cfg, f_s, f_t, p_s, p_t, mask_s, mask_t, matches, randn, t, max_cond):
    """3D/models/pipeline.py:182-216 -> dict(conf_matrix_pred, R, t, noised, src_warped, conf_matrix_gt_hat).  With float64 weights and
    features it is the float64 evaluation the parity tests measure ill-conditioned entries against (positions, the noise draw and the
    Procrustes inputs stay float32 as in the reference)."""
    P, N, _ = f_s.shape
    M = f_t.shape[1]
    a_s, a_t, pe_s, pe_t = coarse_transformer(W, cfg, f_s, f_t, p_s, p_t, mask_s, mask_t, max_cond)
    conf = orc.match_head(W, cfg, a_s, a_t, pe_s, pe_t, mask_s, mask_t, prefix="coarse_matching.")
    R, tt, _, _, _, _ = orc.procrustes(conf.float(), p_s, p_t, mask_s, mask_t, cfg["sample_rate"], max_cond)
    noised = gt_noising(match_matrix(matches, P, N, M), randn, t)
    warped, _ = orc.warp_from_matrix(W, cfg, noised.clone(), p_s, p_t, mask_s, mask_t, max_cond, "3dmatch")
    d_s, d_t, qe_s, qe_t = orc.denoiser(W, cfg, f_s, f_t, warped, p_t.float(), mask_s, mask_t)
    hat = orc.match_head(W, cfg, d_s, d_t, qe_s, qe_t, mask_s, mask_t)
    return dict(conf_matrix_pred=conf, R_s2t_pred=R, t_s2t_pred=tt, noised=noised, src_warped=warped, conf_matrix_gt_hat=hat)


# ------------------------------------------------------------------------------------------------------------------------------
# backward of the matching head's loss, written out (the golden vectors are torch autograd through the reference's own functions)
# ------------------------------------------------------------------------------------------------------------------------------
def focal_loss_backward(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0):
    """d loss / d conf of 3D/models/loss.py:311-314 (clamp :299 passes no gradient outside its range)"""
    pos, neg = conf_gt == 1, conf_gt == 0
    npos, nneg = pos.sum().clamp(min=1), neg.sum().clamp(min=1)
    pw = pos_w if pos.any() else 0.0
    nw = neg_w if neg.any() else 0.0
    c = conf
    live = (c >= 1e-6) & (c <= 1 - 1e-6)
    dpos = -alpha * (-gamma * (1 - c) ** (gamma - 1) * c.log() + (1 - c) ** gamma / c) * (pw / npos)
    dneg = -alpha * (gamma * c ** (gamma - 1) * (1 - c).log() - c ** gamma / (1 - c)) * (nw / nneg)
    g = torch.where(pos, dpos, torch.where(neg, dneg, torch.zeros_like(c)))
    return torch.where(live, g, torch.zeros_like(c))


def sinkhorn_backward(scores, alpha, iters, src_mask, tgt_mask, grad_conf):
    """backward of conf = exp(log_optimal_transport(scores, alpha, iters, masks))[:, :-1, :-1] (3D/models/matching.py:61-93, 213-214) by the
    adjoint recurrences of the iteration (csrc/train.hip states them): -> grad_scores [B,N,M], grad_alpha (0-d)"""
    B, N, M = scores.shape
    dt = scores.dtype
    a = torch.as_tensor(alpha).to(dt)
    rows, cols = src_mask.sum(1, keepdim=True), tgt_mask.sum(1, keepdim=True)
    Z = torch.full((B, N + 1, M + 1), 0.0, dtype=dt)
    Z[:, :N, :M] = scores
    Z[:, :N, M] = a
    Z[:, N, :] = a
    norm = -(rows + cols).log().to(dt)
    log_mu = torch.cat([norm.expand(B, N), cols.log().to(dt) + norm], 1)
    log_nu = torch.cat([norm.expand(B, M), rows.log().to(dt) + norm], 1)
    us, vs = [], [torch.zeros_like(log_nu)]
    for _ in range(iters):
        us.append(log_mu - torch.logsumexp(Z + vs[-1][:, None, :], dim=2))
        vs.append(log_nu - torch.logsumexp(Z + us[-1][:, :, None], dim=1))
    D = torch.zeros_like(Z)
    D[:, :N, :M] = (Z[:, :N, :M] + us[-1][:, :N, None] + vs[-1][:, None, :M] - norm[:, :, None]).exp() * grad_conf
    gZ = D.clone()
    vb = D.sum(1)
    for t in range(iters, 0, -1):
        Pv = (Z + us[t - 1][:, :, None] + vs[t][:, None, :] - log_nu[:, None, :]).exp()
        ub = (D.sum(2) if t == iters else 0) - (vb[:, None, :] * Pv).sum(2)
        gZ = gZ - vb[:, None, :] * Pv
        Pu = (Z + vs[t - 1][:, None, :] + us[t - 1][:, :, None] - log_mu[:, :, None]).exp()
        gZ = gZ - ub[:, :, None] * Pu
        vb = -(ub[:, :, None] * Pu).sum(1)
    g_alpha = gZ[:, N, :].sum() + gZ[:, :N, M].sum()
    return gZ[:, :N, :M].contiguous(), g_alpha
