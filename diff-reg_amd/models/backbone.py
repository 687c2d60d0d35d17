"""models.backbone -- drop-in for the reference's KPFCN backbone, coarse phase (3D/models/backbone.py:6-158), on the HIP
ops of diffreg_hip.

The module tree reproduces the reference's parameter names (encoder_blocks.{i}.KPConv.{weights,kernel_points},
.unary1/.unary2/.unary_shortcut.mlp.weight, decoder_blocks.{i}.mlp.weight, coarse_out / coarse_in / fine_out), so the
`backbone.*` part of a Diff-Reg checkpoint loads unchanged; the kernel point dispositions come from the checkpoint
(the reference regenerates them from kernels/dispositions/*.ply, 3D/models/blocks.py:199-212).  forward(batch,
phase='coarse') returns the coarse features exactly as the reference does -- under .train() with gradients enabled as the
head of an autograd graph whose backward runs on the library's kernels too (diffreg_hip/backbone_autograd.py); other phases are
not accelerated (the reference's own fine branch is commented out upstream, backbone.py:161-180).
"""
import torch
import torch.nn as nn


def _get(cfg, k):
    return cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)


class KPConv(nn.Module):                        # parameters of 3D/models/blocks.py:120-192
    def __init__(self, K, cin, cout):
        super().__init__()
        self.weights = nn.Parameter(torch.zeros(K, cin, cout))
        self.kernel_points = nn.Parameter(torch.zeros(K, 3), requires_grad=False)


class BiasBlock(nn.Module):                     # BatchNormBlock with use_bn = False: a bias per channel (blocks.py:430-446); with use_bn it is an
    def __init__(self, dim):                    # InstanceNorm1d without parameters and this module is not created
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(dim))


class UnaryBlock(nn.Module):                    # blocks.py:455-484
    def __init__(self, cin, cout, use_bn=True):
        super().__init__()
        self.mlp = nn.Linear(cin, cout, bias=False)
        if not use_bn:
            self.batch_norm = BiasBlock(cout)


class SimpleBlock(nn.Module):                   # blocks.py:518-572
    def __init__(self, K, cin, cout, use_bn=True):
        super().__init__()
        self.KPConv = KPConv(K, cin, cout // 2)
        if not use_bn:
            self.batch_norm = BiasBlock(cout // 2)


class ResnetBottleneckBlock(nn.Module):         # blocks.py:575-660
    def __init__(self, K, cin, cout, use_bn=True):
        super().__init__()
        self.unary1 = UnaryBlock(cin, cout // 4, use_bn) if cin != cout // 4 else nn.Identity()
        self.KPConv = KPConv(K, cout // 4, cout // 4)
        if not use_bn:
            self.batch_norm_conv = BiasBlock(cout // 4)
        self.unary2 = UnaryBlock(cout // 4, cout, use_bn)
        self.unary_shortcut = UnaryBlock(cin, cout, use_bn) if cin != cout else nn.Identity()


class NearestUpsampleBlock(nn.Module):          # blocks.py:676-691 (no parameters)
    pass


class KPFCN(nn.Module):
    def __init__(self, config):
        super().__init__()
        arch = list(_get(config, "architecture"))
        K = _get(config, "num_kernel_points")
        self.arch = arch
        self.cfg = dict(num_layers=_get(config, "num_layers"), in_points_dim=3, first_feats_dim=_get(config, "first_feats_dim"),
                        first_subsampling_dl=_get(config, "first_subsampling_dl"), in_feats_dim=_get(config, "in_feats_dim"),
                        conv_radius=_get(config, "conv_radius"), num_kernel_points=K, KP_extent=_get(config, "KP_extent"),
                        coarse_feature_dim=_get(config, "coarse_feature_dim"), KP_influence=_get(config, "KP_influence"),
                        aggregation_mode=_get(config, "aggregation_mode"), use_batch_norm=bool(_get(config, "use_batch_norm")))
        if self.cfg["KP_influence"] not in ("constant", "linear", "gaussian") or self.cfg["aggregation_mode"] not in ("sum", "closest"):
            raise ValueError("KP_influence / aggregation_mode: %r / %r" % (self.cfg["KP_influence"], self.cfg["aggregation_mode"]))   # (blocks.py:321, 329)
        if _get(config, "deformable") or any("deformable" in b for b in arch):
            raise NotImplementedError("deformable KPConv (offset convolutions + their regulariser, blocks.py:214-286) is not accelerated: no shipped "
                                      "configuration selects it")
        use_bn = self.cfg["use_batch_norm"]
        # ---- the reference's construction order (backbone.py:13-112), parameters only --------------------------------
        layer, in_dim, out_dim = 0, _get(config, "in_feats_dim"), _get(config, "first_feats_dim")
        self.encoder_blocks = nn.ModuleList()
        skip_dims, start = [], 0
        for bi, block in enumerate(arch):
            if any(t in block for t in ("pool", "strided", "upsample", "global")):
                skip_dims.append(in_dim)
            if "upsample" in block:
                start = bi
                break
            self.encoder_blocks.append(SimpleBlock(K, in_dim, out_dim, use_bn) if "simple" in block else ResnetBottleneckBlock(K, in_dim, out_dim, use_bn))
            in_dim = out_dim // 2 if "simple" in block else out_dim
            if "pool" in block or "strided" in block:
                layer += 1; out_dim *= 2
        cdim = _get(config, "coarse_feature_dim")
        self.coarse_out = nn.Conv1d(in_dim // 2, cdim, kernel_size=1, bias=True)
        self.coarse_in = nn.Conv1d(cdim, in_dim // 2, kernel_size=1, bias=True)
        self.decoder_blocks = nn.ModuleList()
        for di, block in enumerate(arch[start:]):
            if di > 0 and "upsample" in arch[start + di - 1]:
                in_dim += skip_dims[layer]
            self.decoder_blocks.append(NearestUpsampleBlock() if "upsample" in block else UnaryBlock(in_dim, out_dim, use_bn))
            in_dim = out_dim
            if "upsample" in block:
                layer -= 1; out_dim //= 2
        self.fine_out = nn.Conv1d(out_dim, _get(config, "fine_feature_dim"), kernel_size=1, bias=True)
        self._engine = None
        self._engine_key = None

    def _get_engine(self, device):
        from diffreg_hip.backbone import KPFCNEngine
        key = (str(device), tuple(p._version for p in self.parameters()))
        if self._engine is None or self._engine_key != key:
            self._engine = KPFCNEngine(self.state_dict(), arch=self.arch, cfg=self.cfg, device=device)
            self._engine_key = key
        return self._engine

    def forward(self, batch, phase="encode"):
        if phase != "coarse":
            raise NotImplementedError("only phase='coarse' exists (the reference's other branches are commented out, backbone.py:161-180)")
        if self.training and torch.is_grad_enabled():
            # training: the same kernels as a chain of autograd Functions with backward kernels (diffreg_hip/backbone_autograd.py):
            # gradients reach every parameter of the coarse phase
            from diffreg_hip.backbone_autograd import kpfcn_coarse
            return kpfcn_coarse(self, batch, arch=self.arch, cfg=self.cfg)
        dev = batch["points"][0].device
        with torch.no_grad():
            return self._get_engine(dev).forward(batch)
