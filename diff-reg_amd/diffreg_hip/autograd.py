"""torch.autograd wrappers of the kernels that have a backward: the matching heads (Sinkhorn read-out, focal loss), the
GeometryAttentionLayer with its attention fused in both directions, the weighted Procrustes fit (closed-form adjoint on the device), the L1 motion
term, the row scatter of split_feats; the KPFCN backbone's chain lives in backbone_autograd.py.  Forward and backward both run in libdiffreg_hip:
`Pipeline.forward_train` + `MatchMotionLoss.forward_train` assemble a training step whose .backward() reaches every parameter the reference trains.

    conf = sinkhorn_conf(sim_matrix, bin_score, iters, src_mask, tgt_mask)        # = exp(log_optimal_transport(...))[:, :-1, :-1]
    loss = focal_loss(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0)   # compute_correspondence_loss, sinkhorn form
    loss.backward()                                                                # -> sim_matrix.grad, bin_score.grad
"""
import weakref

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
    def forward(ctx, conf, conf_gt, alpha, gamma, pos_w, neg_w, match_type="sinkhorn"):
        ctx.save_for_backward(conf.detach(), conf_gt)
        # the dual-softmax form (loss.py:301-307) is the positive term alone: the sinkhorn form's backward with neg_w = 0
        ctx.hp = (alpha, gamma, pos_w, neg_w if match_type == "sinkhorn" else 0.0)
        return lib.focal_loss(conf.detach(), conf_gt, None, alpha, gamma, pos_w, neg_w, match_type)

    @staticmethod
    def backward(ctx, grad_loss):
        conf, conf_gt = ctx.saved_tensors
        return lib.focal_loss_backward(conf, conf_gt, *ctx.hp) * grad_loss, None, None, None, None, None, None


class _MatchingHead(torch.autograd.Function):
    """Matching.forward, sinkhorn branch, disentangled rotary code (3D/models/matching.py:164-216): src_proj on both sides (quirk Q1), rotary,
    / sqrt(C), similarity, mask, Sinkhorn read-out.  Backward: dr_sinkhorn_backward_f32, then the similarity / rotary / projection transposes
    on the library's GEMM (dr_linear_f32) and dr_rotary_f32."""

    @staticmethod
    def forward(ctx, src_feats, tgt_feats, weight, bin_score, cs, ss, ct, st, src_mask, tgt_mask, iters):
        B, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        sf, tf, W = src_feats.detach().float().contiguous(), tgt_feats.detach().float().contiguous(), weight.detach().float().contiguous()
        cs, ss, ct, st = (t_.detach().float().contiguous() for t_ in (cs, ss, ct, st))
        spre, tpre = lib.linear(sf.reshape(B * N, C), W), lib.linear(tf.reshape(B * M, C), W)
        a = lib.rotary(spre, cs, ss, scale=1.0 / C ** 0.5).view(B, N, C)
        b = lib.rotary(tpre, ct, st, scale=1.0 / C ** 0.5).view(B, M, C)
        sim = lib.bmm_nt(a, b)
        if src_mask is not None:
            sim = sim.masked_fill(~(src_mask[:, :, None] & tgt_mask[:, None, :]), float("-inf"))
        conf = lib.sinkhorn(sim, bin_score.detach().float().reshape(1), iters, src_mask, tgt_mask)
        ctx.save_for_backward(sf, tf, W, a, b, sim, bin_score.detach(), cs, ss, ct, st, spre, tpre)
        ctx.iters, ctx.masks = iters, (src_mask, tgt_mask)
        return conf

    @staticmethod
    def backward(ctx, grad_conf):
        sf, tf, W, a, b, sim, bin_score, cs, ss, ct, st, spre, tpre = ctx.saved_tensors
        sm, tm = ctx.masks
        B, N, C = sf.shape
        M = tf.shape[1]
        if sm is None:
            sm = torch.ones(B, N, dtype=torch.bool, device=sf.device)
            tm = torch.ones(B, M, dtype=torch.bool, device=sf.device)
        gs, ga = lib.sinkhorn_backward(sim, bin_score, ctx.iters, sm, tm, grad_conf)                  # d loss / d sim  [B,N,M]
        tr = lambda x: x.transpose(-1, -2).contiguous()
        g_a = lib.bmm_nt(gs, tr(b))                                                                   # gs b    [B,N,C]
        g_b = lib.bmm_nt(tr(gs), tr(a))                                                               # gs^T a  [B,M,C]
        g_sp = lib.rotary(g_a.reshape(B * N, C), cs, ss, inverse=True, scale=1.0 / C ** 0.5)          # back through / sqrt(C) and the rotary code
        g_tp = lib.rotary(g_b.reshape(B * M, C), ct, st, inverse=True, scale=1.0 / C ** 0.5)
        Wt = tr(W)
        g_src = lib.linear(g_sp, Wt).view(B, N, C)                                                     # g W
        g_tgt = lib.linear(g_tp, Wt).view(B, M, C)
        pad4 = lambda x: torch.nn.functional.pad(x, (0, (-x.shape[1]) % 4))                            # (the GEMM wants K % 4 == 0)
        g_W = lib.linear(pad4(tr(g_sp)), pad4(tr(sf.reshape(B * N, C)))) + lib.linear(pad4(tr(g_tp)), pad4(tr(tf.reshape(B * M, C))))   # g^T x, both sides
        gcs = gss = gct = gst = None        # (position codes: constants, as in the reference)
        return g_src, g_tgt, g_W, ga.reshape(bin_score.shape).to(bin_score.dtype), gcs, gss, gct, gst, None, None, None


class _ScatterRows(torch.autograd.Function):
    """dst[dst_index[i]] = src[src_index[i]] into a zero tensor of n_dst rows (Pipeline.split_feats, 3D/models/pipeline.py:350-379); backward: the
    same kernel with the index lists swapped (the lists are injective: every destination row has one source)"""

    @staticmethod
    def forward(ctx, src, src_index, dst_index, n_dst, status):
        srcd = src.detach().float().contiguous()
        dst = torch.zeros(n_dst, srcd.shape[1], device=srcd.device)
        lib.scatter_rows(srcd, src_index, dst_index, dst, validate=False, status=status)
        ctx.save_for_backward(src_index, dst_index)
        ctx.n_src = srcd.shape[0]
        return dst

    @staticmethod
    def backward(ctx, g):
        si, di = ctx.saved_tensors
        gs = torch.zeros(ctx.n_src, g.shape[1], device=g.device)
        lib.scatter_rows(g.contiguous().float(), di, si, gs, validate=False)
        return gs, None, None, None, None


def scatter_rows(src, src_index, dst_index, n_dst, status=None):
    """differentiable lib.scatter_rows into a fresh zero tensor [n_dst, C]"""
    return _ScatterRows.apply(src, src_index, dst_index, int(n_dst), status)


def _mm(a, b):
    """a [R,K] @ b[N,K]^T on dr_linear_f32 (K padded to a multiple of 4: the kernel's vector width)"""
    a, b = a.contiguous(), b.contiguous()
    pad = (-a.shape[1]) % 4
    if pad:
        a, b = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad))
    return lib.linear(a, b)


_LAYER_KEYS = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight", "mlp.2.weight", "norm1.weight", "norm1.bias",
               "norm2.weight", "norm2.bias")


def _layer_params(layer):
    """the ten parameters of a GeometryAttentionLayer in _LAYER_KEYS order, by attribute path (`mlp` is an nn.Sequential: mlp[0], mlp[2]) --
    dict(layer.named_parameters()) walks the module tree on every call: 0.4 ms of a training step"""
    return (layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight, layer.merge.weight, layer.mlp[0].weight, layer.mlp[2].weight,
            layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias)


class _GeometryAttentionLayer(torch.autograd.Function):
    """GeometryAttentionLayer.forward (3D/models/transformero.py:43-96, rotary code) and its backward: the attention itself is FUSED both ways
    (forward = the inference kernels, dr_attention_f32; backward = dr_attention_backward_f32, flash-style: per-query log-sum-exp + delta, then dQ by
    query blocks and dK | dV by key blocks on the f32-input MFMA -- no [B,H,L,S] matrix exists in either direction); LayerNorm / ReLU / rotary
    backward kernels (csrc/train.hip); every projection and its weight gradient on dr_linear_f32."""

    #: True (default): forward and backward are ONE library call each (dr_attention_layer_train_forward_f32 / dr_attention_layer_backward_f32:
    #: the same kernels launched from C++ -- a step is ~1 900 launches, and driving them from Python cost more host time than the kernels run).
    #: False: the per-op form below (one library call per kernel), kept as the readable statement of the backward and for A/B tests.
    fused = True

    @staticmethod
    def forward(ctx, x, source, cx, sx, cy, sy, x_mask, source_mask, H, Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2):
        B, L, C = x.shape
        S = source.shape[1]
        d = C // H
        det = lambda t: t.detach().float().contiguous()
        if _GeometryAttentionLayer.fused:
            x3, s3 = det(x), det(source)
            wts = [det(t) for t in (Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2)]
            tabs = [det(t) for t in (cx, sx, cy, sy)]
            out, saved = lib.attention_layer_train_forward(wts, C, H, x3, s3, *tabs, x_mask, source_mask)
            ctx.save_for_backward(x3, s3, saved, *tabs, *wts)
            ctx.dims = (B, L, S, C, H, d, 1.0 / d ** 0.5)
            ctx.masks = (x_mask, source_mask)
            ctx.is_fused = True
            return out
        ctx.is_fused = False
        x2, s2 = det(x).reshape(B * L, C), det(source).reshape(B * S, C)
        Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2 = map(det, (Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2))
        cx, sx, cy, sy = map(det, (cx, sx, cy, sy))
        qpre, kpre = lib.linear(x2, Wq), lib.linear(s2, Wk)
        qw, kw = lib.rotary(qpre, cx, sx), lib.rotary(kpre, cy, sy)
        vw = lib.linear(s2, Wv)
        # fused attention in the token layout (head h in columns h d ..): no [B,H,L,S] matrix, no head permutes
        o2 = lib.attention(qw.view(B, L, C), kw.view(B, S, C), vw.view(B, S, C), H, x_mask, source_mask).view(B * L, C)
        scale = 1.0 / d ** 0.5
        m_pre = lib.linear(o2, Wm)
        m, st1 = lib.layernorm(m_pre, g1, b1)
        cat = torch.cat([x2, m], 1)
        h = lib.linear(cat, W0, epilogue=1)
        f_pre = lib.linear(h, W2)
        f, st2 = lib.layernorm(f_pre, g2, b2)
        ctx.save_for_backward(x2, s2, cx, sx, cy, sy, Wq, Wk, Wv, Wm, W0, W2, g1, g2, qw, kw, vw, o2, m_pre, st1, cat, h, f_pre, st2)
        ctx.dims = (B, L, S, C, H, d, scale)
        ctx.masks = (x_mask, source_mask)
        return (x2 + f).view(B, L, C)

    @staticmethod
    def backward(ctx, ge):
        if ctx.is_fused:
            x3, s3, saved, cx, sx, cy, sy, *wts = ctx.saved_tensors
            B, L, S, C, H, d, scale = ctx.dims
            xm, sm_ = ctx.masks
            gx, gs, gw = lib.attention_layer_backward(wts, C, H, x3, s3, cx, sx, cy, sy, xm, sm_, saved, ge.contiguous().float())
            return (gx, gs, None, None, None, None, None, None, None, *gw)
        (x2, s2, cx, sx, cy, sy, Wq, Wk, Wv, Wm, W0, W2, g1, g2, qw, kw, vw, o2, m_pre, st1, cat, h, f_pre, st2) = ctx.saved_tensors
        B, L, S, C, H, d, scale = ctx.dims
        tr = lambda t: t.transpose(-1, -2).contiguous()
        ge = ge.contiguous().float().reshape(B * L, C)
        g_fpre, gg2, gb2 = lib.layernorm_backward(f_pre, g2, st2, ge)
        g_h = lib.relu_backward(h, _mm(g_fpre, tr(W2)))
        gW2 = _mm(tr(g_fpre), tr(h))
        g_cat = _mm(g_h, tr(W0))
        gW0 = _mm(tr(g_h), tr(cat))
        g_x = ge + g_cat[:, :C]
        g_mpre, gg1, gb1 = lib.layernorm_backward(m_pre, g1, st1, g_cat[:, C:].contiguous())
        g_o2 = _mm(g_mpre, tr(Wm))
        gWm = _mm(tr(g_mpre), tr(o2))
        xm, sm_ = ctx.masks
        g_qw, g_kw, g_vw = lib.attention_backward(qw.view(B, L, C), kw.view(B, S, C), vw.view(B, S, C), o2.view(B, L, C), g_o2.view(B, L, C), H, xm, sm_)
        g_qw, g_kw, g_vw = g_qw.view(B * L, C), g_kw.view(B * S, C), g_vw.view(B * S, C)
        g_qpre = lib.rotary(g_qw, cx, sx, inverse=True)
        g_kpre = lib.rotary(g_kw, cy, sy, inverse=True)
        gcx = gsx = gcy = gsy = None        # (position codes are constants of the graph: the reference detaches them, position_encoding.py:83-84)
        g_x = g_x + _mm(g_qpre, tr(Wq))
        g_s = _mm(g_kpre, tr(Wk)) + _mm(g_vw, tr(Wv))
        gWq, gWk, gWv = _mm(tr(g_qpre), tr(x2)), _mm(tr(g_kpre), tr(s2)), _mm(tr(g_vw), tr(s2))
        return (g_x.view(B, L, C), g_s.view(B, S, C), gcx, gsx, gcy, gsy, None, None, None, gWq, gWk, gWv, gWm, gW0, gW2, gg1, gb1, gg2, gb2)


_TABLES = {}        # id(position-code tensor) -> (weak reference to it, its version counter, (cos, sin)): dropped when the tensor dies


def _tables(pe):
    """a position code [B,N,C,2] (VolumetricPositionEncoding.forward) or a (cos, sin) pair of half tables [B*N, C/2] -> the pair.  The pair of a code
    tensor is built once and kept while that tensor lives unmodified: every layer call and the head of a forward pass ask for the same one (two strided
    copies per request were 200 small launches of a training step)."""
    if isinstance(pe, (tuple, list)):
        return pe
    key = id(pe)
    hit = _TABLES.get(key)
    if hit is not None and hit[0]() is pe and hit[1] == pe._version:
        return hit[2]
    from models.position_encoding import half_tables
    pair = half_tables(pe)
    _TABLES[key] = (weakref.ref(pe, lambda _r, k=key: _TABLES.pop(k, None)), pe._version, pair)
    return pair


def geometry_attention_layer(layer, x, source, x_pe, source_pe, x_mask=None, source_mask=None):
    """differentiable GeometryAttentionLayer.forward for a `models.transformero.GeometryAttentionLayer` module (its parameters receive gradients;
    position codes are constants of the graph, as in the reference: position_encoding.py:83-84 detaches them)"""
    cx, sx = _tables(x_pe)
    cy, sy = _tables(source_pe)
    return _GeometryAttentionLayer.apply(x, source, cx, sx, cy, sy, x_mask, source_mask, layer.nhead, *_layer_params(layer))


class _Procrustes(torch.autograd.Function):
    """SoftProcrustesLayer.forward (3D/models/procrustes.py:17-93) with its gradient into the confidence matrix.  Forward = dr_procrustes_f32 (top-K
    selection, weighted Kabsch, fp64 3 x 3 SVD on the device).  Backward: the K selected confidences are the weights of the fit; their gradient is
    dr_procrustes_backward_f32 -- the closed-form adjoint of the fit (SVD adjoint for 3 x 3 matrices) in float64 on the device, one workgroup
    per pair, replacing the reference's autograd path through `Sxy.cpu().double().svd()` (:35-36)."""

    @staticmethod
    def forward(ctx, conf, src_pcd, tgt_pcd, src_mask, tgt_mask, sample_rate, max_cond, use_mask_len=False):
        R, t, Rf, tf, cond, ok, idx = lib.procrustes(conf.detach().float(), src_pcd, tgt_pcd, src_mask, tgt_mask, sample_rate, max_cond,
                                                     use_mask_len=use_mask_len, want_topk=True)
        ctx.save_for_backward(conf.detach(), src_pcd, tgt_pcd, idx, ok)
        ctx.entry_max = None
        if use_mask_len:                 # 4D (models/procrustes.py:61-76): K from the mask sums, weights beyond a pair's own count are zeroed
            ctx.entry_max = (torch.maximum(src_mask.sum(1), tgt_mask.sum(1)).float() * sample_rate).int()
        ctx.mark_non_differentiable(cond, ok)
        return R, t, Rf, tf, cond, ok

    @staticmethod
    def backward(ctx, gR, gt, gRf, gtf, _gc, _gk):
        conf, ps, pt, idx, ok = ctx.saved_tensors
        B = conf.shape[0]
        okf = ok.view(B, 1, 1).to(gR.dtype)
        gR_eff, gt_eff = gR + gRf * okf, gt + gtf * okf                    # R_forwd = R where the condition gate passes, identity elsewhere (:85-90)
        g_conf = lib.procrustes_backward(conf, ps, pt, idx, gR_eff, gt_eff, k_count=ctx.entry_max)
        return g_conf, None, None, None, None, None, None, None


def procrustes_fit(layer, conf, src_pcd, tgt_pcd, src_mask, tgt_mask):
    """differentiable models.procrustes.SoftProcrustesLayer.forward -> R, t, R_forwd, t_forwd, condition, solution_mask"""
    return _Procrustes.apply(conf, src_pcd.float().contiguous(), tgt_pcd.float().contiguous(), src_mask, tgt_mask, float(layer.sample_rate),
                             float(layer.max_condition_num), bool(getattr(layer, "use_mask_len", False)))


class _MotionL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_pcd, R_pred, t_pred, R_gt, t_gt, overlap_mask, flow):
        ctx.save_for_backward(s_pcd, R_pred.detach(), t_pred.detach(), R_gt, t_gt, overlap_mask)
        ctx.flow = flow
        return lib.motion_l1(s_pcd, R_pred.detach(), t_pred.detach(), R_gt, t_gt, overlap_mask, flow)

    @staticmethod
    def backward(ctx, g):
        s_pcd, Rp, tp, Rg, tg, om = ctx.saved_tensors
        gR, gt = lib.motion_l1_backward(s_pcd, Rp, tp, Rg, tg, om, ctx.flow)
        return None, gR * g, gt * g, None, None, None, None


def motion_l1(s_pcd, R_pred, t_pred, R_gt, t_gt, overlap_mask, flow=None):
    """differentiable L1 motion term of ge_coarse_loss (3D/models/loss.py:108-128) in (R_pred, t_pred)"""
    return _MotionL1.apply(s_pcd, R_pred, t_pred, R_gt, t_gt, overlap_mask, flow)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# The configuration branches no shipped yaml selects (pe_type 'sinusoidal', entangled = True, match_type 'dual_softmax'), differentiable: the
# per-kernel form of the layer / head above with the position code entering where that branch puts it.  Same kernels, one library call each;
# gradients pinned to the reference's autograd by tests/test_train_branches_gpu.py (oracle/make_golden_train_branches.py).
# ---------------------------------------------------------------------------------------------------------------------------------------------
class _Rotary(torch.autograd.Function):
    """embed_rotary (position_encoding.py:25-35) on [B,N,C] features with half tables: the entangled rotary form rotates the features ONCE in front of
    the layers (transformero.py:238-239); backward = the transposed rotation"""

    @staticmethod
    def forward(ctx, x, cos, sin):
        B, N, C = x.shape
        ctx.save_for_backward(cos, sin)
        return lib.rotary(x.detach().float().reshape(B * N, C).contiguous(), cos, sin).view(B, N, C)

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        B, N, C = g.shape
        return lib.rotary(g.contiguous().float().reshape(B * N, C), cos, sin, inverse=True).view(B, N, C), None, None


class _GeometryAttentionLayerG(torch.autograd.Function):
    """GeometryAttentionLayer.forward in the forms of transformero.py:50-57 / 246-252: q = W_q (x + x_add), k = W_k (source + s_add), v = W_v source
    (x_add / s_add: the sinusoidal code, or None), the rotary code on q and k only when tables are given (None: the entangled forms call the layers
    without a code).  The per-kernel backward of _GeometryAttentionLayer with those inputs."""

    @staticmethod
    def forward(ctx, x, source, x_add, s_add, cx, sx, cy, sy, x_mask, source_mask, H, Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2):
        B, L, C = x.shape
        S = source.shape[1]
        d = C // H
        det = lambda t: t.detach().float().contiguous()
        x2, s2 = det(x).reshape(B * L, C), det(source).reshape(B * S, C)
        qin = x2 if x_add is None else x2 + det(x_add).reshape(B * L, C)
        kin = s2 if s_add is None else s2 + det(s_add).reshape(B * S, C)
        Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2 = map(det, (Wq, Wk, Wv, Wm, W0, W2, g1, b1, g2, b2))
        rot = cx is not None
        qw, kw = lib.linear(qin, Wq), lib.linear(kin, Wk)
        if rot:
            cx, sx, cy, sy = map(det, (cx, sx, cy, sy))
            qw, kw = lib.rotary(qw, cx, sx), lib.rotary(kw, cy, sy)
        vw = lib.linear(s2, Wv)
        o2 = lib.attention(qw.view(B, L, C), kw.view(B, S, C), vw.view(B, S, C), H, x_mask, source_mask).view(B * L, C)
        m_pre = lib.linear(o2, Wm)
        m, st1 = lib.layernorm(m_pre, g1, b1)
        cat = torch.cat([x2, m], 1)
        h = lib.linear(cat, W0, epilogue=1)
        f_pre = lib.linear(h, W2)
        f, st2 = lib.layernorm(f_pre, g2, b2)
        tabs = (cx, sx, cy, sy) if rot else tuple(torch.empty(0, device=x2.device) for _ in range(4))
        ctx.save_for_backward(qin, kin, s2, *tabs, Wq, Wk, Wv, Wm, W0, W2, g1, g2, qw, kw, vw, o2, m_pre, st1, cat, h, f_pre, st2)
        ctx.dims = (B, L, S, C, H, rot)
        ctx.masks = (x_mask, source_mask)
        return (x2 + f).view(B, L, C)

    @staticmethod
    def backward(ctx, ge):
        (qin, kin, s2, cx, sx, cy, sy, Wq, Wk, Wv, Wm, W0, W2, g1, g2, qw, kw, vw, o2, m_pre, st1, cat, h, f_pre, st2) = ctx.saved_tensors
        B, L, S, C, H, rot = ctx.dims
        tr = lambda t: t.transpose(-1, -2).contiguous()
        ge = ge.contiguous().float().reshape(B * L, C)
        g_fpre, gg2, gb2 = lib.layernorm_backward(f_pre, g2, st2, ge)
        g_h = lib.relu_backward(h, _mm(g_fpre, tr(W2)))
        gW2 = _mm(tr(g_fpre), tr(h))
        g_cat = _mm(g_h, tr(W0))
        gW0 = _mm(tr(g_h), tr(cat))
        g_x = ge + g_cat[:, :C]
        g_mpre, gg1, gb1 = lib.layernorm_backward(m_pre, g1, st1, g_cat[:, C:].contiguous())
        g_o2 = _mm(g_mpre, tr(Wm))
        gWm = _mm(tr(g_mpre), tr(o2))
        xm, sm_ = ctx.masks
        g_qw, g_kw, g_vw = lib.attention_backward(qw.view(B, L, C), kw.view(B, S, C), vw.view(B, S, C), o2.view(B, L, C), g_o2.view(B, L, C), H, xm, sm_)
        g_qw, g_kw, g_vw = g_qw.view(B * L, C), g_kw.view(B * S, C), g_vw.view(B * S, C)
        g_qpre = lib.rotary(g_qw, cx, sx, inverse=True) if rot else g_qw.contiguous()
        g_kpre = lib.rotary(g_kw, cy, sy, inverse=True) if rot else g_kw.contiguous()
        g_x = g_x + _mm(g_qpre, tr(Wq))
        g_s = _mm(g_kpre, tr(Wk)) + _mm(g_vw, tr(Wv))
        gWq, gWk, gWv = _mm(tr(g_qpre), tr(qin)), _mm(tr(g_kpre), tr(kin)), _mm(tr(g_vw), tr(s2))
        # (the position codes -- x_add / s_add and the tables -- are constants of the graph: position_encoding.py:83-84 detaches them)
        return (g_x.view(B, L, C), g_s.view(B, S, C), None, None, None, None, None, None, None, None, None, gWq, gWk, gWv, gWm, gW0, gW2, gg1, gb1, gg2, gb2)


def geometry_attention_layer_form(layer, x, source, x_pe, source_pe, x_mask=None, source_mask=None):
    """differentiable GeometryAttentionLayer.forward in the form the module's own pe_type selects; x_pe / source_pe = None: the call of the entangled
    forms (no code inside the layer).  The shipped form (rotary code given) takes the fused library call of geometry_attention_layer."""
    pe_type = getattr(layer, "pe_type", "rotary")
    if pe_type == "rotary" and x_pe is not None:
        return geometry_attention_layer(layer, x, source, x_pe, source_pe, x_mask, source_mask)
    w = _layer_params(layer)
    if x_pe is None:
        return _GeometryAttentionLayerG.apply(x, source, None, None, None, None, None, None, x_mask, source_mask, layer.nhead, *w)
    if pe_type == "sinusoidal":
        return _GeometryAttentionLayerG.apply(x, source, x_pe, source_pe, None, None, None, None, x_mask, source_mask, layer.nhead, *w)
    raise KeyError(pe_type)


class _MatchingHeadG(torch.autograd.Function):
    """Matching.forward (3D/models/matching.py:164-216) in every form: src_proj on both sides (quirk Q1); the position code by `embed` -- 'rotary'
    (tables), 'add' (the sinusoidal code, position_encoding.py:43-44) or 'none' (entangled, matching.py:181); / sqrt(C); similarity; the read-out:
    Sinkhorn (bin_score) or dual softmax (temperature).  Backward: the read-out's kernel, then the similarity / code / projection transposes."""

    @staticmethod
    def forward(ctx, src_feats, tgt_feats, weight, bin_score, embed, pe_a, pe_b, pe_c, pe_d, src_mask, tgt_mask, readout, iters, temperature):
        B, N, C = src_feats.shape
        M = tgt_feats.shape[1]
        sf, tf, W = src_feats.detach().float().contiguous(), tgt_feats.detach().float().contiguous(), weight.detach().float().contiguous()
        spre, tpre = lib.linear(sf.reshape(B * N, C), W), lib.linear(tf.reshape(B * M, C), W)
        sc = 1.0 / C ** 0.5
        if embed == "rotary":
            cs, ss, ct, st = (t_.detach().float().contiguous() for t_ in (pe_a, pe_b, pe_c, pe_d))
            a, b = lib.rotary(spre, cs, ss, scale=sc).view(B, N, C), lib.rotary(tpre, ct, st, scale=sc).view(B, M, C)
        else:
            cs = ss = ct = st = torch.empty(0, device=sf.device)
            if embed == "add":
                spre, tpre = spre + pe_a.detach().float().reshape(B * N, C), tpre + pe_b.detach().float().reshape(B * M, C)
            a, b = (spre * sc).view(B, N, C), (tpre * sc).view(B, M, C)
        sim = lib.bmm_nt(a, b)
        if readout == "dual_softmax":
            conf = lib.dual_softmax(sim, temperature, src_mask, tgt_mask)
            bs = torch.empty(0, device=sf.device)
        else:
            if src_mask is not None:
                sim = sim.masked_fill(~(src_mask[:, :, None] & tgt_mask[:, None, :]), float("-inf"))
            bs = bin_score.detach()
            conf = lib.sinkhorn(sim, bs.float().reshape(1), iters, src_mask, tgt_mask)
        ctx.save_for_backward(sf, tf, W, a, b, sim, bs, cs, ss, ct, st)
        ctx.cfg = (embed, readout, iters, temperature, bin_score is not None)
        ctx.masks = (src_mask, tgt_mask)
        return conf

    @staticmethod
    def backward(ctx, grad_conf):
        sf, tf, W, a, b, sim, bs, cs, ss, ct, st = ctx.saved_tensors
        embed, readout, iters, temperature, has_bin = ctx.cfg
        sm, tm = ctx.masks
        B, N, C = sf.shape
        M = tf.shape[1]
        ga = None
        if readout == "dual_softmax":
            gs = lib.dual_softmax_backward(sim, temperature, sm, tm, grad_conf)
        else:
            if sm is None:
                sm = torch.ones(B, N, dtype=torch.bool, device=sf.device)
                tm = torch.ones(B, M, dtype=torch.bool, device=sf.device)
            gs, ga = lib.sinkhorn_backward(sim, bs, iters, sm, tm, grad_conf)
            ga = ga.reshape(bs.shape).to(bs.dtype)
        tr = lambda x: x.transpose(-1, -2).contiguous()
        g_a = lib.bmm_nt(gs, tr(b))
        g_b = lib.bmm_nt(tr(gs), tr(a))
        sc = 1.0 / C ** 0.5
        if embed == "rotary":
            g_sp = lib.rotary(g_a.reshape(B * N, C), cs, ss, inverse=True, scale=sc)
            g_tp = lib.rotary(g_b.reshape(B * M, C), ct, st, inverse=True, scale=sc)
        else:
            g_sp, g_tp = (g_a * sc).reshape(B * N, C).contiguous(), (g_b * sc).reshape(B * M, C).contiguous()
        Wt = tr(W)
        g_src, g_tgt = lib.linear(g_sp, Wt).view(B, N, C), lib.linear(g_tp, Wt).view(B, M, C)
        g_W = _mm(tr(g_sp), tr(sf.reshape(B * N, C))) + _mm(tr(g_tp), tr(tf.reshape(B * M, C)))
        return g_src, g_tgt, g_W, (ga if has_bin else None), None, None, None, None, None, None, None, None, None, None


def matching_head_form(m, src_feats, tgt_feats, src_pe, tgt_pe, src_mask, tgt_mask, pe_type):
    """differentiable models.matching.Matching.forward for the module `m` in the form its configuration selects (entangled, pe_type, match_type)"""
    readout = m.match_type
    bin_score = m.bin_score if readout == "sinkhorn" else None
    iters = int(getattr(m, "skh_iters", 0) or 0)
    temperature = float(getattr(m, "temperature", 1.0) or 1.0)
    W = m.src_proj.weight
    if m.entangled:
        return _MatchingHeadG.apply(src_feats, tgt_feats, W, bin_score, "none", None, None, None, None, src_mask, tgt_mask, readout, iters, temperature)
    if pe_type == "rotary":
        if readout == "sinkhorn":
            return matching_head(src_feats, tgt_feats, W, bin_score, src_pe, tgt_pe, src_mask, tgt_mask, iters)          # the shipped form
        cs, ss = _tables(src_pe)
        ct, st = _tables(tgt_pe)
        return _MatchingHeadG.apply(src_feats, tgt_feats, W, bin_score, "rotary", cs, ss, ct, st, src_mask, tgt_mask, readout, iters, temperature)
    if pe_type == "sinusoidal":
        return _MatchingHeadG.apply(src_feats, tgt_feats, W, bin_score, "add", src_pe, tgt_pe, None, None, src_mask, tgt_mask, readout, iters, temperature)
    raise KeyError(pe_type)


def _embed_features(pe_type, feats, pe):
    """VolPE.embed_pos on the features (the entangled forms, transformero.py:238-239), differentiable in the features"""
    if pe_type == "rotary":
        c, s_ = _tables(pe)
        return _Rotary.apply(feats, c, s_)
    if pe_type == "sinusoidal":
        return feats + pe
    raise KeyError(pe_type)


def _layers_of(tr, s, t, src_pe, tgt_pe, src_mask, tgt_mask, positioning=None):
    """the self / cross layers of a RepositioningTransformer in the form its configuration selects (transformero.py:170-254), differentiable:
    disentangled -- every layer call receives the codes (rotary tables / the sinusoidal code); entangled -- the code is put into the features once
    and the layers are called without one (positioning layers are skipped there, transformero.py:252).  positioning(s, t, src_pe, tgt_pe) ->
    the new source code of a positioning layer (disentangled forms), or None where the transformer must not contain one."""
    ent = bool(tr.entangled)
    if ent:
        s, t = _embed_features(tr.pe_type, s, src_pe), _embed_features(tr.pe_type, t, tgt_pe)
    code = (lambda pe_: None) if ent else (lambda pe_: pe_)
    for layer, name in zip(tr.layers, tr.layer_types):
        if name == "self":
            s = geometry_attention_layer_form(layer, s, s, code(src_pe), code(src_pe), src_mask, src_mask)
            t = geometry_attention_layer_form(layer, t, t, code(tgt_pe), code(tgt_pe), tgt_mask, tgt_mask)
        elif name == "cross":
            s = geometry_attention_layer_form(layer, s, t, code(src_pe), code(tgt_pe), src_mask, tgt_mask)
            t = geometry_attention_layer_form(layer, t, s, code(tgt_pe), code(src_pe), tgt_mask, src_mask)          # the updated src (quirk Q11)
        elif ent:
            continue
        elif positioning is not None:
            src_pe = positioning(layer, s, t, src_pe, tgt_pe)
        else:
            raise NotImplementedError("positioning layers (Procrustes inside the transformer) have no backward")
    return s, t, src_pe, tgt_pe


def coarse_branch(pipeline, src_feats, tgt_feats, s_pcd, t_pcd, src_mask, tgt_mask):
    """The non-denoising half of the training forward, differentiable (3D/models/pipeline.py:184-196): coarse_transformer -- self, cross, the
    positioning layer (its Matching + Procrustes fit re-pose the source for the position code, which the reference DETACHES,
    position_encoding.py:83-84: a constant of the graph), self, cross -- then coarse_matching and the final soft_procrustes
    -> (conf_matrix_pred, R_s2t_pred, t_s2t_pred); (R, t) back-propagate into conf_matrix_pred (the L1 motion term of 4DMatch's training).
    Every configuration form (pe_type, entangled, match_type); positioning_type 'procrustes' (the others re-pose with data the training graph does
    not carry here)."""
    tr = pipeline.coarse_transformer
    pe = tr.positional_encoding
    with torch.no_grad():
        src_pe, tgt_pe = pe(s_pcd), pe(t_pcd)

    def positioning(layer, s, t, spe, tpe):                          # (transformero.py:183-203): no gradient leaves it
        if tr.positioning_type != "procrustes":
            raise NotImplementedError("forward_train: positioning_type %r" % tr.positioning_type)
        with torch.no_grad():
            conf, _ = layer[0](s.detach(), t.detach(), spe, tpe, src_mask, tgt_mask, {}, pe_type=tr.pe_type)
            _, _, Rf, tf, _, _ = layer[1](conf, s_pcd, t_pcd, src_mask, tgt_mask)
            return pe((torch.matmul(Rf, s_pcd.transpose(1, 2)) + tf).transpose(1, 2))
    s, t, src_pe, tgt_pe = _layers_of(tr, src_feats, tgt_feats, src_pe, tgt_pe, src_mask, tgt_mask, positioning)
    conf = matching_head_form(pipeline.coarse_matching, s, t, src_pe, tgt_pe, src_mask, tgt_mask, tr.pe_type)
    R, tt, _, _, _, _ = procrustes_fit(pipeline.soft_procrustes, conf, s_pcd, t_pcd, src_mask, tgt_mask)
    return conf, R, tt


def denoising_branch(pipeline, src_feats, tgt_feats, src_pcd_wrapped, tgt_pcd_wrapped, src_mask, tgt_mask):
    """The denoising half of the training forward, differentiable (3D/models/pipeline.py:209-212): denoising_transformer (six self / cross
    GeometryAttentionLayers on the position code of the warped source) + denoising_coarse_matching -> conf_matrix_gt_hat.  Gradients reach every
    parameter of the two modules and the backbone features; the warped points (from the noised ground-truth matrix: no parameters) are constants,
    exactly as in the reference's graph.  Every configuration form (pe_type, entangled, match_type)."""
    tr = pipeline.denoising_transformer
    with torch.no_grad():
        src_pe, tgt_pe = tr.positional_encoding(src_pcd_wrapped), tr.positional_encoding(tgt_pcd_wrapped)
    s, t, src_pe, tgt_pe = _layers_of(tr, src_feats, tgt_feats, src_pe, tgt_pe, src_mask, tgt_mask)
    return matching_head_form(pipeline.denoising_coarse_matching, s, t, src_pe, tgt_pe, src_mask, tgt_mask, tr.pe_type)


def matching_head(src_feats, tgt_feats, weight, bin_score, src_pe, tgt_pe, src_mask, tgt_mask, iters):
    """differentiable Matching.forward (sinkhorn, rotary): src_pe / tgt_pe = position codes [B,N,C,2] or (cos, sin) half-table pairs"""
    cs, ss = _tables(src_pe)
    ct, st = _tables(tgt_pe)
    return _MatchingHead.apply(src_feats, tgt_feats, weight, bin_score, cs, ss, ct, st, src_mask, tgt_mask, int(iters))


def sinkhorn_conf(scores, bin_score, iters, src_mask=None, tgt_mask=None):
    """differentiable exp(log_optimal_transport(scores, bin_score, iters, masks))[:, :-1, :-1] (3D/models/matching.py:207-216); masked entries of
    `scores` are filled with -inf here, as the reference does before the call"""
    return _SinkhornConf.apply(scores, bin_score, int(iters), src_mask, tgt_mask)


def focal_loss(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0, match_type="sinkhorn"):
    """differentiable compute_correspondence_loss (3D/models/loss.py:273-314): the sinkhorn form, or the dual-softmax form (positive term only)"""
    return _FocalLoss.apply(conf, conf_gt, float(alpha), float(gamma), float(pos_w), float(neg_w), match_type)
