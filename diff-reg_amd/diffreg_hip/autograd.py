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


def sinkhorn_conf(scores, bin_score, iters, src_mask=None, tgt_mask=None):
    """differentiable exp(log_optimal_transport(scores, bin_score, iters, masks))[:, :-1, :-1] (3D/models/matching.py:207-216); masked entries of
    `scores` are filled with -inf here, as the reference does before the call"""
    return _SinkhornConf.apply(scores, bin_score, int(iters), src_mask, tgt_mask)


def focal_loss(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0):
    """differentiable compute_correspondence_loss, sinkhorn form (3D/models/loss.py:273-314)"""
    return _FocalLoss.apply(conf, conf_gt, float(alpha), float(gamma), float(pos_w), float(neg_w))
