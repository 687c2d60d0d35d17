"""The KPFCN backbone's coarse phase with gradients (second half): KPFCN.forward(batch, phase='coarse')
(3D/models/backbone.py:122-158) as a chain of torch.autograd.Functions whose forward AND backward run on libdiffreg_hip -- the same
forward ops as diffreg_hip.backbone.KPFCNEngine (one gather kernel + one GEMM per KPConv, the column-statistics normalisation fused with
LeakyReLU and the residual sum, the pools) and their backward kernels (csrc/backbone_bwd.hip); every nn.Linear half is a product on the
library's GEMM.  Gradients reach all 55 parameter tensors the reference trains in this phase and the input features.

    coarse = kpfcn_coarse(module, batch)        # module: models.backbone.KPFCN (or any module with the reference's parameter names)
    loss(coarse).backward()                     # -> module.encoder_blocks[..].KPConv.weights.grad, ....mlp.weight.grad, coarse_out.*.grad
"""
import torch

from . import lib


def _mm(a, b):
    """a [R,K] @ b [N,K]^T on dr_linear_f32 (K padded to a multiple of 4: the kernel's vector width)"""
    a, b = a.contiguous(), b.contiguous()
    pad = (-a.shape[1]) % 4
    if pad:
        a, b = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad))
    return lib.linear(a, b)


_tr = lambda t: t.transpose(0, 1).contiguous()

_WG_CHUNK = 1024


def _weight_grad(g, x):
    """g^T x over the points, [R, Cout] x [R, Cin] -> [Cout, Cin].  One MFMA chain over thousands of points accumulates float32 rounding along
    its whole length (4e-4 of the tensor's maximum at 8 000 points without a normalisation in front); beyond 2 chunks the contraction is TWO-LEVEL --
    chunks of 1 024 points as one batched launch, their partial products summed in float64 -- which is what a blocked CPU sgemm does implicitly."""
    R = g.shape[0]
    if R <= 2 * _WG_CHUNK:
        return _mm(_tr(g), _tr(x))
    n = (R + _WG_CHUNK - 1) // _WG_CHUNK
    pad = n * _WG_CHUNK - R
    if pad:
        g, x = torch.nn.functional.pad(g, (0, 0, 0, pad)), torch.nn.functional.pad(x, (0, 0, 0, pad))
    gT = g.view(n, _WG_CHUNK, g.shape[1]).transpose(1, 2)                 # [n, Cout, chunk]
    xT = x.view(n, _WG_CHUNK, x.shape[1]).transpose(1, 2)                 # [n, Cin, chunk]
    return lib.bmm_nt(gT, xT).double().sum(0).float()


class _Linear(torch.autograd.Function):
    """y = x W^T (+ bias): UnaryBlock.mlp, coarse_out (a 1 x 1 Conv1d) and the single GEMM of a KPConv"""

    @staticmethod
    def forward(ctx, x, W, bias):
        xd, Wd = x.detach().float().contiguous(), W.detach().float().contiguous()
        ctx.save_for_backward(xd, Wd)
        ctx.has_bias = bias is not None
        return lib.linear_ex(xd, Wd, bias=bias.detach().float().contiguous() if bias is not None else None)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous().float()
        gx = _mm(g, _tr(W)) if ctx.needs_input_grad[0] else None          # g W
        gW = _weight_grad(g, x) if ctx.needs_input_grad[1] else None         # g^T x
        gb = g.double().sum(0).float() if ctx.has_bias else None           # (a sum over all points: float64, like the weight gradient's second level)
        return gx, gW, gb


class _KPGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, q, s, idx, kp, extent, influence="linear", aggregation="sum"):
        xd = x.detach().float().contiguous()
        ctx.save_for_backward(xd, q, s, idx, kp)
        ctx.cfg = (extent, influence, aggregation)
        return lib.kpconv_gather(q, s, idx, xd, kp, extent, influence, aggregation)

    @staticmethod
    def backward(ctx, g):
        x, q, s, idx, kp = ctx.saved_tensors
        extent, influence, aggregation = ctx.cfg
        return lib.kpconv_gather_backward(q, s, idx, x, kp, extent, g.contiguous().float(), influence, aggregation), None, None, None, None, None, None, None


class _BiasAct(torch.autograd.Function):
    """act( (a + bias_a) + [(b + bias_b) | b | 0] ): the BatchNormBlock of use_batch_norm = False (a bias per channel, blocks.py:445-446) + LeakyReLU(0.1)
    + the residual sum.  Forward = dr_norm_apply_f32 with (mean, rstd) = (-bias, 1); backward: the LeakyReLU's slope read off the output's sign (torch
    element-wise glue) and column sums for the biases."""

    @staticmethod
    def forward(ctx, a, bias_a, b, bias_b, activate):
        ad = a.detach().float().contiguous()
        bd = b.detach().float().contiguous() if b is not None else None
        one = torch.ones_like(bias_a.detach())
        sb = ((-bias_b.detach()).contiguous(), one) if bias_b is not None else None
        out = lib.norm_apply(ad, ((-bias_a.detach()).contiguous(), one), bd, sb, activate=activate)
        ctx.save_for_backward(out)
        ctx.cfg = (b is not None, bias_b is not None, activate)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        has_b, has_bb, act = ctx.cfg
        g = g.contiguous().float()
        if act:
            g = torch.where(out > 0, g, g * 0.1)
        gs = g.double().sum(0).float()
        return g, gs, (g if has_b else None), (gs if has_bb else None), None


class _Norm(torch.autograd.Function):
    """act( norm(a) + [norm(b) | b | 0] ): BatchNormBlock = InstanceNorm1d over the points (blocks.py:430-446) + LeakyReLU(0.1) + the residual sum"""

    @staticmethod
    def forward(ctx, a, b, norm_b, activate):
        ad = a.detach().float().contiguous()
        bd = b.detach().float().contiguous() if b is not None else None
        sa = lib.col_stats(ad)
        sb = lib.col_stats(bd) if (bd is not None and norm_b) else None
        out = lib.norm_apply(ad, sa, bd, sb, activate=activate)
        ctx.save_for_backward(ad, sa[0], sa[1], bd if bd is not None else ad.new_empty(0), *(sb if sb is not None else (ad.new_empty(0), ad.new_empty(0))), out)
        ctx.cfg = (b is not None, sb is not None, activate)
        return out

    @staticmethod
    def backward(ctx, g):
        a, ma, ra, b, mb, rb, out = ctx.saved_tensors
        has_b, nb, act = ctx.cfg
        ga, gb = lib.norm_backward(g.contiguous().float(), out, a, (ma, ra), b if has_b else None, (mb, rb) if nb else None, activate=act)
        return ga, gb, None, None


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, inds, first_only):
        xd = x.detach().float().contiguous()
        ctx.save_for_backward(xd, inds)
        ctx.first_only = first_only
        return lib.gather_pool(xd, inds, first_only=first_only)

    @staticmethod
    def backward(ctx, g):
        x, inds = ctx.saved_tensors
        return lib.gather_pool_backward(x, inds, g.contiguous().float(), first_only=ctx.first_only), None, None


def _w2(weights):
    """KPConv.weights [K, Cin, Cout] -> the [Cout, ceil4(K Cin)] operand of the single GEMM (differentiable torch view ops: glue)"""
    K, Cin, Cout = weights.shape
    w2 = weights.permute(2, 0, 1).reshape(Cout, K * Cin)
    pad = (-K * Cin) % 4
    return torch.nn.functional.pad(w2, (0, pad)) if pad else w2


def kpfcn_coarse(module, batch, arch=None, cfg=None):
    """differentiable KPFCN.forward(batch, phase='coarse') for a module with the reference's parameter names -> coarse features
    [N_coarse, coarse_feature_dim]; the kernel points and the clouds are constants of the graph."""
    from .synth import KPFCN_ARCH, KPFCN_CFG
    arch = list(arch if arch is not None else getattr(module, "arch", KPFCN_ARCH))
    cfg = dict(cfg if cfg is not None else getattr(module, "cfg", KPFCN_CFG))
    P = dict(module.named_parameters())
    dev = next(module.parameters()).device
    pts = [p.to(dev, torch.float32).contiguous() for p in batch["points"]]
    nb = [i.to(dev, torch.int64).contiguous() for i in batch["neighbors"]]
    pools = [i.to(dev, torch.int64).contiguous() for i in batch["pools"]]
    ups = [i.to(dev, torch.int64).contiguous() for i in batch["upsamples"]]
    x = batch["features"].to(dev, torch.float32).contiguous()
    layer = 0
    r = cfg["first_subsampling_dl"] * cfg["conv_radius"]
    skips, skip_x = [], []

    influence, aggregation, use_bn = cfg.get("KP_influence", "linear"), cfg.get("aggregation_mode", "sum"), bool(cfg.get("use_batch_norm", True))

    def kpconv(pre, q, s, idx, y, extent):
        wf = _KPGather.apply(y, q, s, idx, P[pre + "KPConv.kernel_points"].detach().float().contiguous(), extent, influence, aggregation)
        return _Linear.apply(wf, _w2(P[pre + "KPConv.weights"]), None)

    def norm(a, key_a, b=None, key_b=None, norm_b=False):
        """the BatchNormBlock(s) + LeakyReLU + residual sum of a block: InstanceNorm statistics, or the bias form of use_batch_norm = False"""
        if use_bn:
            return _Norm.apply(a, b, norm_b, True)
        return _BiasAct.apply(a, P[key_a], b, P[key_b] if (b is not None and norm_b) else None, True)

    for bi, block in enumerate(arch):
        if any(t in block for t in ("pool", "strided", "upsample", "global")):
            skips.append(bi)
        if "upsample" in block:
            break
        if bi in skips:
            skip_x.append(x)
        pre = "encoder_blocks.%d." % bi
        extent = r * cfg["KP_extent"] / cfg["conv_radius"]
        strided = "strided" in block
        q, s, idx = (pts[layer + 1], pts[layer], pools[layer]) if strided else (pts[layer], pts[layer], nb[layer])
        if block == "simple":
            x = norm(kpconv(pre, q, s, idx, x, extent), pre + "batch_norm.bias")
        else:
            feats = x
            y = feats
            if (pre + "unary1.mlp.weight") in P:
                y = norm(_Linear.apply(feats, P[pre + "unary1.mlp.weight"], None), pre + "unary1.batch_norm.bias")
            y = norm(kpconv(pre, q, s, idx, y, extent), pre + "batch_norm_conv.bias")
            y = _Linear.apply(y, P[pre + "unary2.mlp.weight"], None)                       # unary2: norm only (no_relu)
            sc = _Pool.apply(feats, idx, False) if strided else feats
            if (pre + "unary_shortcut.mlp.weight") in P:
                sc = _Linear.apply(sc, P[pre + "unary_shortcut.mlp.weight"], None)
                x = norm(y, pre + "unary2.batch_norm.bias", sc, pre + "unary_shortcut.batch_norm.bias", True)   # lrelu(norm(y) + norm(sc))
            else:
                x = norm(y, pre + "unary2.batch_norm.bias", sc, None, False)                 # lrelu(norm(y) + feats)
        if "pool" in block or "strided" in block:
            layer += 1; r *= 2
    x = _Pool.apply(x, ups[layer - 1], True)                                                 # nearest upsample
    x = torch.cat([x, skip_x.pop()], 1)
    x = norm(_Linear.apply(x, P["decoder_blocks.1.mlp.weight"], None), "decoder_blocks.1.batch_norm.bias")
    return _Linear.apply(x, P["coarse_out.weight"][:, :, 0], P["coarse_out.bias"])
