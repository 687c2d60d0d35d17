"""Mint vectors for the KPConv / BatchNormBlock options NO shipped yaml selects, by RUNNING THE REFERENCE backbone (build container only; needs
/root/reference):

    python oracle/make_golden_kpfcn_variants.py      # writes tests/golden/kpfcn_variants.npz

  KP_influence 'constant' / 'gaussian'   3D/models/blocks.py:304-321 (radius_gaussian :36-44)
  aggregation_mode 'closest'             blocks.py:324-326
  use_batch_norm = False                 blocks.py:430-446: every BatchNormBlock is a bias per channel
(deformable KPConv -- offset convolutions and their regulariser, blocks.py:214-286 -- is not built: NotImplementedError in models/backbone.py.)
Per variant: the reference's coarse output (every second row) and, under autograd with loss = sum(coarse * G), 256 sampled entries of the gradient
of every parameter of the coarse phase (+ the same entries from a float64 evaluation of the restatement, oracle/kpfcn_oracle.py, which is asserted
against the reference here before anything is written).  Inputs and weights: diffreg_hip.synth (the batch and weights of make_golden_kpfcn.py;
the BatchNormBlock biases of the use_batch_norm = False variants from synth.make_kpfcn_bn_biases)."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "kpfcn_variants.npz")
TREE = "/root/reference/Diff-Reg-3dmatch"
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)

VARIANTS = {"gauss_sum_bn": ("gaussian", "sum", True), "const_closest_bn": ("constant", "closest", True), "linear_sum_nobn": ("linear", "sum", False),
            "gauss_closest_nobn": ("gaussian", "closest", False)}


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def main():
    import torch
    from diffreg_hip import synth
    sys.modules["open3d"] = MagicMock()
    os.chdir(TREE)
    sys.path.insert(0, TREE)
    from models.backbone import KPFCN
    from oracle import kpfcn_oracle as ko
    used = ("encoder_blocks.", "decoder_blocks.1.", "coarse_out.")
    b = synth.make_kpfcn_batch()
    tb = dict(points=[torch.from_numpy(p) for p in b["points"]], neighbors=[torch.from_numpy(p) for p in b["neighbors"]],
              pools=[torch.from_numpy(p) for p in b["pools"]], upsamples=[torch.from_numpy(p) for p in b["upsamples"]],
              stack_lengths=[torch.tensor(l) for l in b["stack_lengths"]], features=torch.from_numpy(b["features"]))
    res = {}
    for tag, (influence, aggregation, use_bn) in VARIANTS.items():
        cfg = AttrDict(dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), final_feats_dim=32, deform_radius=5.0, KP_influence=influence,
                            aggregation_mode=aggregation, fixed_kernel_points="center", use_batch_norm=use_bn, batch_norm_momentum=0.02, deformable=False,
                            modulated=False, fine_feature_dim=264))
        torch.manual_seed(0)
        np.random.seed(0)
        net = KPFCN(cfg).eval()
        ref_sd = net.state_dict()
        kp = {k: v.numpy().copy() for k, v in ref_sd.items() if k.endswith("kernel_points")}
        sd = synth.make_kpfcn_weights(kp)
        if not use_bn:
            sd.update(synth.make_kpfcn_bn_biases([(k, tuple(v.shape)) for k, v in ref_sd.items() if k.startswith(used) and ".batch_norm" in k and k.endswith(".bias")]))
        missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(not m.startswith(used) for m in missing), (unexpected, [m for m in missing if m.startswith(used)])
        feats_in = tb["features"].clone()
        out = net(dict(tb, features=feats_in), phase="coarse")
        G = torch.from_numpy(synth.hash_normal(77, 1, tuple(out.shape)).astype(np.float32))
        (out * G).sum().backward()
        named = dict(net.named_parameters())
        ocfg = dict(synth.KPFCN_CFG, KP_influence=influence, aggregation_mode=aggregation, use_batch_norm=use_bn)
        grad_keys = [k for k in sorted(named) if k.startswith(used) and named[k].grad is not None and not k.endswith("kernel_points")]
        # the restatement against the reference: values and gradients (float32), then its float64 evaluation as the yardstick
        t32 = {k: torch.from_numpy(v).clone().requires_grad_(k in grad_keys) for k, v in sd.items()}
        mine = ko.kpfcn_coarse(t32, tb, cfg=ocfg)
        err = float((mine - out).abs().max())
        (mine * G).sum().backward()
        worst = max(float((t32[k].grad - named[k].grad).abs().max() / (named[k].grad.abs().max() + 1e-30)) for k in grad_keys)
        print(tag, "coarse", tuple(out.shape), "abs max %.3f" % float(out.abs().max()), "| oracle - reference %.2e, worst relative gradient deviation %.2e over %d tensors"
              % (err, worst, len(grad_keys)))
        assert err < 2e-4 * max(1.0, float(out.abs().max())) and worst < 2e-3
        t64 = {k: (torch.from_numpy(v).double() if v.dtype.kind == "f" else torch.from_numpy(v)).clone().requires_grad_(k in grad_keys) for k, v in sd.items()}
        tb64 = dict(tb, points=[p.double() for p in tb["points"]], features=tb["features"].double())
        (ko.kpfcn_coarse(t64, tb64, cfg=ocfg) * G.double()).sum().backward()
        res[tag + ":coarse"] = out.detach().numpy()[::2]
        for i, k in enumerate(grad_keys):
            g_ = named[k].grad.detach().reshape(-1)
            idx = (synth.hash_u01(78, i + 1, 256) * g_.numel()).astype(np.int64)
            res["%s:gidx:%s" % (tag, k)] = idx
            res["%s:gval:%s" % (tag, k)] = g_[torch.from_numpy(idx)].numpy()
            res["%s:gmax:%s" % (tag, k)] = np.array(float(g_.abs().max()))
            res["%s:g64val:%s" % (tag, k)] = t64[k].grad.reshape(-1)[torch.from_numpy(idx)].numpy().astype(np.float32)
        res[tag + ":grad_keys"] = np.array(grad_keys)
        res.update({"%s:kp:%s" % (tag, k): v for k, v in kp.items()})
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(res), "arrays")


if __name__ == "__main__":
    main()
