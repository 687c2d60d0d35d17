"""torch.autograd wrappers of the kernels that have a backward (first backward kernels): the matching head's
Sinkhorn read-out and its focal loss.  Forward and backward both run in libdiffreg_hip; the rest of the training graph (projections,
attention layers, Procrustes) has no backward kernels yet, so these are the differentiable tail of the model, not a trainer.

    conf = sinkhorn_conf(sim_matrix, bin_score, iters, src_mask, tgt_mask)        # = exp(log_optimal_transport(...))[:, :-1, :-1]
    loss = focal_loss(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0)   # compute_correspondence_loss, sinkhorn form
    loss.backward()                                                                # -> sim_matrix.grad, bin_score.grad
"""
import torch

from . import lib


class _SinkhornConf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, bin_score, iters, src_mask, tgt_mask):
        masked = scores.masked_fill(~(src_mask[:, :, None] & tgt_mask[:, None, :]), float("-inf")) if src_mask is not None else scores
        conf = lib.sinkhorn(masked.detach().float(), bin_score.detach().float().reshape(1), iters, src_mask, tgt_mask)
        ctx.save_for_backward(masked.detach().float(), bin_score.detach())
        ctx.iters, ctx.masks = iters, (src_mask, tgt_mask)
        return conf

    @staticmethod
    def backward(ctx, grad_conf):
        masked, bin_score = ctx.saved_tensors
        sm, tm = ctx.masks
        if sm is None:
            sm = torch.ones(masked.shape[:2], dtype=torch.bool, device=masked.device)
            tm = torch.ones(masked.shape[0], masked.shape[2], dtype=torch.bool, device=masked.device)
        gs, ga = lib.sinkhorn_backward(masked, bin_score, ctx.iters, sm, tm, grad_conf)
        return gs, ga.reshape(bin_score.shape).to(bin_score.dtype), None, None, None


class _FocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conf, conf_gt, alpha, gamma, pos_w, neg_w):
        ctx.save_for_backward(conf.detach(), conf_gt)
        ctx.hp = (alpha, gamma, pos_w, neg_w)
        return lib.focal_loss(conf.detach(), conf_gt, None, alpha, gamma, pos_w, neg_w, "sinkhorn")

    @staticmethod
    def backward(ctx, grad_loss):
        conf, conf_gt = ctx.saved_tensors
        return lib.focal_loss_backward(conf, conf_gt, *ctx.hp) * grad_loss, None, None, None, None, None


class _MatchingHead(torch.autograd.Function):
    """Matching.forward, sinkhorn branch, disentangled rotary code (3D/models/matching.py:164-216): src_proj on both sides (quirk Q1), rotary,
    / sqrt(C), similarity, mask, Sinkhorn read-out.  Backward: dr_sinkhorn_backward_f32, then the similarity / rotary / projection transposes
    on the library's GEMM (dr_linear_f32) and dr_rotary_f32."""

    @staticmethod
    def forward(ctx, src_feats, tgt_feats, weight, bin_score, cs, ss, ct, st, src_mask, tgt_mask, iters):
        B, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        sf, tf, W = src_feats.detach().float().contiguous(), tgt_feats.detach().float().contiguous(), weight.detach().float().contiguous()
        a = lib.linear(sf.reshape(B * N, C), W, epilogue=2, cos=cs, sin=ss, rot_C=C).view(B, N, C) / C ** 0.5
        b = lib.linear(tf.reshape(B * M, C), W, epilogue=2, cos=ct, sin=st, rot_C=C).view(B, M, C) / C ** 0.5
        sim = torch.stack([lib.linear(a[i], b[i]) for i in range(B)])
        if src_mask is not None:
            sim = sim.masked_fill(~(src_mask[:, :, None] & tgt_mask[:, None, :]), float("-inf"))
        conf = lib.sinkhorn(sim, bin_score.detach().float().reshape(1), iters, src_mask, tgt_mask)
        ctx.save_for_backward(sf, tf, W, a, b, sim, bin_score.detach(), cs, ss, ct, st)
        ctx.iters, ctx.masks = iters, (src_mask, tgt_mask)
        return conf

    @staticmethod
    def backward(ctx, grad_conf):
        sf, tf, W, a, b, sim, bin_score, cs, ss, ct, st = ctx.saved_tensors
        sm, tm = ctx.masks
        B, N, C = sf.shape
        M = tf.shape[1]
        if sm is None:
            sm = torch.ones(B, N, dtype=torch.bool, device=sf.device)
            tm = torch.ones(B, M, dtype=torch.bool, device=sf.device)
        gs, ga = lib.sinkhorn_backward(sim, bin_score, ctx.iters, sm, tm, grad_conf)                  # d loss / d sim  [B,N,M]
        tr = lambda x: x.transpose(-1, -2).contiguous()
        g_a = torch.stack([lib.linear(gs[i], tr(b[i])) for i in range(B)])                            # gs b    [B,N,C]
        g_b = torch.stack([lib.linear(tr(gs[i]), tr(a[i])) for i in range(B)])                        # gs^T a  [B,M,C]
        g_sp = lib.rotary(g_a.reshape(B * N, C), cs, ss, inverse=True, scale=1.0 / C ** 0.5)          # back through / sqrt(C) and the rotary code
        g_tp = lib.rotary(g_b.reshape(B * M, C), ct, st, inverse=True, scale=1.0 / C ** 0.5)
        Wt = tr(W)
        g_src = lib.linear(g_sp, Wt).view(B, N, C)                                                     # g W
        g_tgt = lib.linear(g_tp, Wt).view(B, M, C)
        pad4 = lambda x: torch.nn.functional.pad(x, (0, (-x.shape[1]) % 4))                            # (the GEMM wants K % 4 == 0)
        g_W = lib.linear(pad4(tr(g_sp)), pad4(tr(sf.reshape(B * N, C)))) + lib.linear(pad4(tr(g_tp)), pad4(tr(tf.reshape(B * M, C))))   # g^T x, both sides
        return g_src, g_tgt, g_W, ga.reshape(bin_score.shape).to(bin_score.dtype), None, None, None, None, None, None, None


def matching_head(src_feats, tgt_feats, weight, bin_score, src_pe, tgt_pe, src_mask, tgt_mask, iters):
    """differentiable Matching.forward (sinkhorn, rotary): src_pe / tgt_pe = the position codes [B,N,C,2] of VolumetricPositionEncoding"""
    from models.position_encoding import half_tables
    cs, ss = half_tables(src_pe)
    ct, st = half_tables(tgt_pe)
    return _MatchingHead.apply(src_feats, tgt_feats, weight, bin_score, cs, ss, ct, st, src_mask, tgt_mask, int(iters))


def sinkhorn_conf(scores, bin_score, iters, src_mask=None, tgt_mask=None):
    """differentiable exp(log_optimal_transport(scores, bin_score, iters, masks))[:, :-1, :-1] (3D/models/matching.py:207-216); masked entries of
    `scores` are filled with -inf here, as the reference does before the call"""
    return _SinkhornConf.apply(scores, bin_score, int(iters), src_mask, tgt_mask)


def focal_loss(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0):
    """differentiable compute_correspondence_loss, sinkhorn form (3D/models/loss.py:273-314)"""
    return _FocalLoss.apply(conf, conf_gt, float(alpha), float(gamma), float(pos_w), float(neg_w))
