"""CPU restatement of the KPFCN backbone's coarse phase -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Every function
cites the reference code it follows (3D/ = /root/reference/Diff-Reg-3dmatch/).  Pinned against the reference
itself: oracle/make_golden_kpfcn.py imports models.backbone.KPFCN from the reference tree, runs it on the
synthetic batch of diffreg_hip.synth.make_kpfcn_batch with the hash-generated weights, and stores its output in
tests/golden/kpfcn_coarse.npz; tests/test_oracle_golden.py holds this restatement to it.
"""
import torch
import torch.nn.functional as F

from diffreg_hip.synth import KPFCN_ARCH, KPFCN_CFG


def kpconv(q_pts, s_pts, neighb_inds, x, weights, kernel_points, extent, influence="linear", aggregation="sum"):
    """KPConv.forward, rigid kernel (3D/models/blocks.py:214-360): KP_influence 'constant' / 'linear' (every shipped yaml) / 'gaussian',
    aggregation_mode 'sum' (shipped) / 'closest'.
    q_pts [Nq,3], s_pts [Ns,3], neighb_inds [Nq,H] (shadow index = Ns), x [Ns,Cin], weights [K,Cin,Cout]."""
    s_pad = torch.cat((s_pts, torch.zeros_like(s_pts[:1]) + 1e6), 0)                 # :288
    neighbors = s_pad[neighb_inds] - q_pts.unsqueeze(1)                              # :291-294 [Nq,H,3]
    diff = neighbors.unsqueeze(2) - kernel_points                                    # :304-305 [Nq,H,K,3]
    sq = (diff ** 2).sum(3)                                                          # :308
    if influence == "constant":                                                      # :304-307
        w = torch.ones_like(sq)
    elif influence == "linear":                                                      # :309-312
        w = torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0)
    elif influence == "gaussian":                                                    # :314-318; radius_gaussian :36-44
        w = torch.exp(-sq / (2 * (extent * 0.3) ** 2 + 1e-9))
    else:
        raise ValueError(influence)
    if aggregation == "closest":                                                     # :324-326: only the nearest kernel point of a neighbour
        w = w * F.one_hot(torch.argmin(sq, dim=2), sq.shape[2]).to(w.dtype)
    w = w.transpose(1, 2)                                                            # [Nq,K,H]
    x_pad = torch.cat((x, torch.zeros_like(x[:1])), 0)                               # :369
    nx = x_pad[neighb_inds]                                                          # :372 [Nq,H,Cin]
    wf = torch.matmul(w, nx)                                                         # :375 [Nq,K,Cin]
    out = torch.matmul(wf.permute(1, 0, 2), weights).sum(0)                          # :382-387 [Nq,Cout]
    num = (nx.sum(-1) > 0.0).sum(-1)                                                 # :390-391 neighbours with a positive feature sum
    num = torch.max(num, torch.ones_like(num))
    return out / num.unsqueeze(1)                                                    # :393


def norm_block(x, bias=None):
    """BatchNormBlock with use_bn: nn.InstanceNorm1d over the points of the whole stacked cloud, per channel, no affine
    parameters, biased variance, eps 1e-5 (3D/models/blocks.py:430-446: [N,C] -> [1,C,N] -> InstanceNorm1d); without use_bn
    (bias given): x + bias (:445-446)."""
    if bias is not None:
        return x + bias
    return F.instance_norm(x.t().unsqueeze(0)).squeeze(0).t()


def unary(x, W, relu=True, bias=None):
    """UnaryBlock: bias-free Linear, norm, LeakyReLU(0.1) (3D/models/blocks.py:455-484)"""
    x = norm_block(x @ W.t(), bias)
    return F.leaky_relu(x, 0.1) if relu else x


def max_pool(x, inds):
    """3D/models/blocks.py:71-87 (a zero row stands for the shadow index)"""
    return torch.cat((x, torch.zeros_like(x[:1])), 0)[inds].max(1)[0]


def closest_pool(x, inds):
    """3D/models/blocks.py:56-68"""
    return torch.cat((x, torch.zeros_like(x[:1])), 0)[inds[:, 0]]


def kpfcn_coarse(sd, batch, arch=KPFCN_ARCH, cfg=KPFCN_CFG):
    """KPFCN.forward(batch, phase='coarse') (3D/models/backbone.py:122-158): the encoder, the first upsample + unary of
    the decoder, coarse_out.  sd: state dict (reference names) of torch tensors; batch: torch tensors as in the collate."""
    pts, nb, pools, ups = batch["points"], batch["neighbors"], batch["pools"], batch["upsamples"]
    influence, aggregation = cfg.get("KP_influence", "linear"), cfg.get("aggregation_mode", "sum")
    b = (lambda key: None) if cfg.get("use_batch_norm", True) else (lambda key: sd[key])       # the bias of a BatchNormBlock without use_bn
    x = batch["features"].clone()
    layer, in_dim, out_dim = 0, cfg["in_feats_dim"], cfg["first_feats_dim"]
    r = cfg["first_subsampling_dl"] * cfg["conv_radius"]
    skips, skip_x, bi = [], [], 0
    for bi, block in enumerate(arch):
        if any(t in block for t in ("pool", "strided", "upsample", "global")):
            skips.append(bi)
        if "upsample" in block:
            break
        if bi in skips:
            skip_x.append(x)                                                          # backbone.py:131-132
        pre = "encoder_blocks.%d." % bi
        extent = r * cfg["KP_extent"] / cfg["conv_radius"]                            # blocks.py:530, 587
        strided = "strided" in block
        q, s, idx = (pts[layer + 1], pts[layer], pools[layer]) if strided else (pts[layer], pts[layer], nb[layer])
        kpc = lambda y_: kpconv(q, s, idx, y_, sd[pre + "KPConv.weights"], sd[pre + "KPConv.kernel_points"], extent, influence, aggregation)
        if block == "simple":                                                         # SimpleBlock.forward blocks.py:558-572
            x = F.leaky_relu(norm_block(kpc(x), b(pre + "batch_norm.bias")), 0.1)
        else:                                                                         # ResnetBottleneckBlock.forward blocks.py:630-660
            feats = x
            y = unary(feats, sd[pre + "unary1.mlp.weight"], bias=b(pre + "unary1.batch_norm.bias")) if (pre + "unary1.mlp.weight") in sd else feats
            y = F.leaky_relu(norm_block(kpc(y), b(pre + "batch_norm_conv.bias")), 0.1)
            y = unary(y, sd[pre + "unary2.mlp.weight"], relu=False, bias=b(pre + "unary2.batch_norm.bias"))
            sc = max_pool(feats, idx) if strided else feats
            if (pre + "unary_shortcut.mlp.weight") in sd:
                sc = unary(sc, sd[pre + "unary_shortcut.mlp.weight"], relu=False, bias=b(pre + "unary_shortcut.batch_norm.bias"))
            x = F.leaky_relu(y + sc, 0.1)
        in_dim = out_dim // 2 if "simple" in block else out_dim
        if "pool" in block or "strided" in block:
            layer += 1; r *= 2; out_dim *= 2
    # decoder blocks 0 (nearest_upsample) and 1 (unary on [x | skip]); then coarse_out (backbone.py:149-158)
    x = closest_pool(x, ups[layer - 1])                                               # NearestUpsampleBlock blocks.py:686-687
    x = torch.cat([x, skip_x.pop()], 1)
    x = unary(x, sd["decoder_blocks.1.mlp.weight"], bias=b("decoder_blocks.1.batch_norm.bias"))
    return x @ sd["coarse_out.weight"][:, :, 0].t() + sd["coarse_out.bias"]          # Conv1d(kernel_size=1) on [1,C,N]
