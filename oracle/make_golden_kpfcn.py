"""Mint the KPFCN golden vector by RUNNING THE REFERENCE backbone (build container only; needs /root/reference).

    python oracle/make_golden_kpfcn.py      # writes tests/golden/kpfcn_coarse.npz

models.backbone.KPFCN is imported from the reference tree where it lies (cwd = the tree root so that its kernel
point disposition file resolves); shims: an attribute-dict config, a MagicMock for open3d.  Inputs (the synthetic
stacked cloud with its neighbour / pool / upsample index arrays) and weights come from diffreg_hip.synth; the
fixture stores the reference's kernel points (tiny), its coarse output and a few intermediate block outputs.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
TREE = "/root/reference/Diff-Reg-3dmatch"
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def main():
    import torch
    from diffreg_hip import synth
    sys.modules["open3d"] = MagicMock()
    os.chdir(TREE)
    sys.path.insert(0, TREE)
    from models.backbone import KPFCN                                 # the reference implementation
    cfg = AttrDict(dict(synth.KPFCN_CFG, architecture=list(synth.KPFCN_ARCH), final_feats_dim=32, deform_radius=5.0,
                        KP_influence="linear", aggregation_mode="sum", fixed_kernel_points="center", use_batch_norm=True,
                        batch_norm_momentum=0.02, deformable=False, modulated=False, fine_feature_dim=264))
    torch.manual_seed(0)
    np.random.seed(0)                                                 # the reference rotates its kernel dispositions randomly
    net = KPFCN(cfg).eval()
    ref_sd = net.state_dict()
    kp = {k: v.numpy().copy() for k, v in ref_sd.items() if k.endswith("kernel_points")}
    sd = synth.make_kpfcn_weights(kp)
    for k, v in sd.items():
        assert tuple(ref_sd[k].shape) == tuple(v.shape), (k, ref_sd[k].shape, v.shape)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected
    used = ("encoder_blocks.", "decoder_blocks.1.", "coarse_out.")
    assert all(not m.startswith(used) for m in missing), [m for m in missing if m.startswith(used)]
    b = synth.make_kpfcn_batch()
    tb = dict(points=[torch.from_numpy(p) for p in b["points"]], neighbors=[torch.from_numpy(p) for p in b["neighbors"]],
              pools=[torch.from_numpy(p) for p in b["pools"]], upsamples=[torch.from_numpy(p) for p in b["upsamples"]],
              stack_lengths=[torch.tensor(l) for l in b["stack_lengths"]], features=torch.from_numpy(b["features"]))
    # intermediate outputs of the reference blocks, for localising a mismatch
    inter = {}
    hooks = [net.encoder_blocks[i].register_forward_hook(lambda m, a, o, i=i: inter.__setitem__("enc%d" % i, o.detach().numpy().copy()))
             for i in (0, 1, 2, 10)]
    with torch.no_grad():
        out = net(tb, phase="coarse")
    for h in hooks:
        h.remove()
    # also the restatement, run here against the reference before the fixture is written
    from oracle import kpfcn_oracle as ko
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    mine = ko.kpfcn_coarse(tsd, tb)
    err = (mine - out).abs().max().item()
    print("reference coarse feats", tuple(out.shape), "abs max %.3f" % out.abs().max().item(), "| oracle - reference max abs %.2e" % err)
    assert err < 2e-4 * max(1.0, out.abs().max().item())
    # ---- gradients (row f3): the reference backbone under autograd, loss = sum(coarse * G) with a hash-generated G -> d loss / d every
    # parameter of the coarse phase and d loss / d input features; stored per tensor as its norm and 256 sampled entries (hash indices)
    for p_ in net.parameters():
        p_.grad = None
    feats_in = tb["features"].clone().requires_grad_(True)
    tb_g = dict(tb, features=feats_in)
    out_g = net(tb_g, phase="coarse")
    G = torch.from_numpy(synth.hash_normal(77, 1, tuple(out_g.shape)).astype(np.float32))
    (out_g * G).sum().backward()
    grads = {}
    named = dict(net.named_parameters())
    for k in sorted(named):
        if not k.startswith(used) or named[k].grad is None:
            continue
        g_ = named[k].grad.detach().reshape(-1)
        idx = (synth.hash_u01(78, len(grads) + 1, 256) * g_.numel()).astype(np.int64)
        grads["gnorm:" + k] = np.array(float(g_.double().norm()))
        grads["gidx:" + k] = idx
        grads["gval:" + k] = g_[torch.from_numpy(idx)].numpy()
        grads["gmax:" + k] = np.array(float(g_.abs().max()))
    assert feats_in.grad is None          # (the reference detaches the input features, backbone.py:124: they are constants of its graph)
    # the oracle (torch CPU restatement) under autograd must give the same gradients
    tsd_g = {k: torch.from_numpy(v).clone().requires_grad_(k.startswith(used) and not k.endswith("kernel_points")) for k, v in sd.items()}
    fin2 = tb["features"].clone().requires_grad_(True)
    (ko.kpfcn_coarse(tsd_g, dict(tb, features=fin2)) * G).sum().backward()
    worst = 0.0
    for k in sorted(named):
        if ("gnorm:" + k) in grads:
            worst = max(worst, float((tsd_g[k].grad - named[k].grad).abs().max() / (named[k].grad.abs().max() + 1e-30)))
    print("parameter tensors with gradients:", sum(1 for k in grads if k.startswith("gnorm:")), "| oracle-vs-reference worst relative gradient deviation %.2e" % worst)
    assert worst < 1e-3
    # the same gradients from a FLOAT64 evaluation of the restatement (same sampled entries): the reference's own float32 backward is up to
    # 1.8e-3 of a tensor's largest entry away from it (eleven blocks of InstanceNorm backward over 182 .. 32k points in float32), which is
    # what a float32 implementation can be held to against the reference; against float64 the bar is a plain 1e-4
    t64 = {k: (torch.from_numpy(v).double() if v.dtype.kind == "f" else torch.from_numpy(v)).clone().requires_grad_(k.startswith(used) and not k.endswith("kernel_points"))
           for k, v in sd.items()}
    tb64 = dict(tb, points=[p.double() for p in tb["points"]], features=tb["features"].double())
    (ko.kpfcn_coarse(t64, tb64) * G.double()).sum().backward()
    dev32 = 0.0
    for k in sorted(named):
        if ("gnorm:" + k) in grads:
            g64 = t64[k].grad.reshape(-1)
            grads["g64val:" + k] = g64[torch.from_numpy(grads["gidx:" + k])].numpy()
            grads["g64norm:" + k] = np.array(float(g64.norm()))
            dev32 = max(dev32, float(np.abs(grads["g64val:" + k] - grads["gval:" + k]).max() / grads["gmax:" + k]))
    print("reference float32 gradients vs float64 evaluation: worst sampled deviation %.2e of a tensor's largest entry" % dev32)
    os.makedirs(OUT, exist_ok=True)
    keys = sorted(ref_sd.keys())
    np.savez_compressed(os.path.join(OUT, "kpfcn_coarse.npz"), coarse=out.numpy(),
                        enc0=inter["enc0"][:64], enc1=inter["enc1"][:64], enc2=inter["enc2"][:64], enc10=inter["enc10"][:, :256],
                        # the reference module's parameter names and shapes (state-dict compatibility of models/backbone.py)
                        sd_keys=np.array(keys), sd_shapes=np.array([";".join(map(str, ref_sd[k].shape)) for k in keys]),
                        **{"kp:" + k: v for k, v in kp.items()}, **grads)
    print("wrote", os.path.join(OUT, "kpfcn_coarse.npz"))


if __name__ == "__main__":
    main()
