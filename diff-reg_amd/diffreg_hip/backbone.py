"""KPFCN backbone, coarse phase, on the HIP ops of csrc/backbone.hip.

KPFCNEngine(state_dict) takes the reference's state dict (`backbone.*` names of 3D/models/backbone.py:8-118, with or
without the `backbone.` prefix) and evaluates KPFCN.forward(batch, phase='coarse') (backbone.py:122-158) for a batch
dict as produced by the reference's collate function (points / neighbors / pools / upsamples / features).
Every KPConv is one gather kernel + ONE GEMM over K*Cin (the reference multiplies per kernel point and sums);
BatchNormBlock (an InstanceNorm1d over the points, blocks.py:430-446) is a column-statistics pass fused with the
LeakyReLU / residual sum at application time.  No CPU fallback: device tensors only.
"""
import torch

from . import lib
from .synth import KPFCN_ARCH, KPFCN_CFG


class KPFCNEngine:
    def __init__(self, state, arch=KPFCN_ARCH, cfg=KPFCN_CFG, device="cuda:0"):
        self.arch, self.cfg, self.device = list(arch), dict(cfg), torch.device(device)
        sd = {}
        for k, v in state.items():
            k = k[len("backbone."):] if k.startswith("backbone.") else k
            if k.startswith(("encoder_blocks.", "decoder_blocks.1.", "coarse_out.")):
                sd[k] = v.detach().to(self.device, torch.float32).contiguous()
        self.sd = sd
        # [K, Cin, Cout] -> [Cout, K*Cin (padded to a multiple of 4)]: the B operand of the single GEMM of a KPConv
        self.w2 = {}
        for k, v in sd.items():
            if k.endswith("KPConv.weights"):
                K, Cin, Cout = v.shape
                w2 = v.permute(2, 0, 1).reshape(Cout, K * Cin)
                pad = (-K * Cin) % 4
                if pad:
                    w2 = torch.cat([w2, torch.zeros(Cout, pad, device=self.device)], 1)
                self.w2[k] = w2.contiguous()
        self.coarse_w = sd["coarse_out.weight"][:, :, 0].contiguous()

        self.influence, self.aggregation = self.cfg.get("KP_influence", "linear"), self.cfg.get("aggregation_mode", "sum")
        self.use_bn = bool(self.cfg.get("use_batch_norm", True))

    # ---- blocks -------------------------------------------------------------------------------------------------
    def _kpconv(self, pre, q, s, idx, x, extent):
        wf = lib.kpconv_gather(q, s, idx, x, self.sd[pre + "KPConv.kernel_points"], extent, self.influence, self.aggregation)
        return lib.linear_ex(wf, self.w2[pre + "KPConv.weights"])

    def _stats(self, y, bias_key):
        """the (mean, rstd) a BatchNormBlock applies: the column statistics of the InstanceNorm1d, or -- use_batch_norm = False: x + bias,
        blocks.py:445-446 -- (-bias, 1): (y - mean) rstd is then y + bias exactly"""
        if self.use_bn:
            return lib.col_stats(y)
        return (-self.sd[bias_key]).contiguous(), torch.ones_like(self.sd[bias_key])

    def _unary(self, x, pre, relu=True):
        y = lib.linear_ex(x, self.sd[pre + "mlp.weight"])
        return lib.norm_apply(y, self._stats(y, pre + "batch_norm.bias"), activate=relu)

    @torch.no_grad()
    def forward(self, batch):
        """-> coarse features [N_coarse, coarse_feature_dim] (rows = batch['points'][-2])"""
        cfg, sd = self.cfg, self.sd
        dev = self.device
        pts = [p.to(dev, torch.float32).contiguous() for p in batch["points"]]
        nb = [i.to(dev, torch.int64).contiguous() for i in batch["neighbors"]]
        pools = [i.to(dev, torch.int64).contiguous() for i in batch["pools"]]
        ups = [i.to(dev, torch.int64).contiguous() for i in batch["upsamples"]]
        x = batch["features"].to(dev, torch.float32).contiguous()
        layer, out_dim = 0, cfg["first_feats_dim"]
        r = cfg["first_subsampling_dl"] * cfg["conv_radius"]
        skips, skip_x = [], []
        for bi, block in enumerate(self.arch):
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
                y = self._kpconv(pre, q, s, idx, x, extent)
                x = lib.norm_apply(y, self._stats(y, pre + "batch_norm.bias"))
            else:
                feats = x
                y = self._unary(feats, pre + "unary1.") if (pre + "unary1.mlp.weight") in sd else feats
                y = self._kpconv(pre, q, s, idx, y, extent)
                y = lib.norm_apply(y, self._stats(y, pre + "batch_norm_conv.bias"))
                y = lib.linear_ex(y, sd[pre + "unary2.mlp.weight"])                      # unary2: norm only (no_relu)
                sc = lib.gather_pool(feats, idx) if strided else feats
                if (pre + "unary_shortcut.mlp.weight") in sd:
                    sc = lib.linear_ex(sc, sd[pre + "unary_shortcut.mlp.weight"])
                    x = lib.norm_apply(y, self._stats(y, pre + "unary2.batch_norm.bias"), sc, self._stats(sc, pre + "unary_shortcut.batch_norm.bias"))       # lrelu(norm(y) + norm(sc))
                else:
                    x = lib.norm_apply(y, self._stats(y, pre + "unary2.batch_norm.bias"), sc, None)                    # lrelu(norm(y) + feats)
            if "pool" in block or "strided" in block:
                layer += 1; r *= 2; out_dim *= 2
        x = lib.gather_pool(x, ups[layer - 1], first_only=True)                          # nearest upsample
        x = torch.cat([x, skip_x.pop()], 1)
        x = self._unary(x, "decoder_blocks.1.")
        return lib.linear_ex(x, self.coarse_w, bias=sd["coarse_out.bias"])
