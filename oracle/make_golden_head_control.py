"""A control for the stress head's gradient bound, minted by RUNNING THE REFERENCE's matching head (build container only; needs /root/reference).

    python oracle/make_golden_head_control.py        # writes tests/golden/train_backward_head_control.npz

The stress head (HEAD_GAIN 24) makes matching logits in the thousands: their fp32 ulp is 1.2e-4, so the confidences -- and the focal loss's
gradients -- of two float32 evaluations of the SAME head on the SAME inputs differ at the 1e-4 .. 1e-2 level when only the summation order of
`sim = einsum("bsc,btc->bst")` differs.  This script quantifies that with the reference's own module (3D/models/matching.py:163-216) on the
reference's own layer outputs (tests/golden/train_backward_upstream.npz: branch_out_*32), three ways:
    h64_*      the module in float64 on those float32 inputs                       (the mathematics)
    h32_*      the module as shipped                                                (checked bit-equal to the end-to-end run's head gradient)
    h32perm_*  the module as shipped with the feature PAIRS of src_proj's output rows (and of the rotary code) permuted: the same function of its
               inputs, another summation order in the einsum -- a second float32 evaluation through the reference's code
for conf, the gradient to the two inputs, to src_proj.weight (entries [::6, ::6]) and to bin_score.  tests/test_train_gpu.py holds the device's
head, on the same inputs, to: at least as close to h64 as twice the further of the two float32 evaluations.  Only reference OUTPUTS are stored.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
TREE = "/root/reference/Diff-Reg-3dmatch"
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle.make_golden import ref_config, HEAD_GAIN
    from oracle.make_golden_train import LOSS_CFG
    sys.modules["open3d"] = MagicMock()
    for m in ("easydict", "tensorboardX", "nibabel", "nibabel.quaternions", "cv2"):
        sys.modules.setdefault(m, MagicMock())
    os.chdir(TREE)
    sys.path.insert(0, TREE)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from diffreg_hip import synth
    from models.pipeline import Pipeline
    from models.loss import MatchMotionLoss
    from configs.models import architectures
    from tests.helpers import train_case

    C = synth.VARIANTS["3dmatch"]["C"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    Wnp = dict(synth.make_weights(C, seed=7, head_gain=HEAD_GAIN))
    Wnp.update(synth.make_weights_coarse(C, seed=17, head_gain=HEAD_GAIN))
    up = np.load(os.path.join(GOLD, "train_backward_upstream.npz"))
    fwd = np.load(os.path.join(GOLD, "train_forward.npz"))
    cb = train_case("b1")
    crit = MatchMotionLoss(dict(LOSS_CFG))
    rng = np.random.RandomState(5)
    pairs = rng.permutation(C // 2)
    perm = torch.from_numpy(np.stack([2 * pairs, 2 * pairs + 1], 1).reshape(-1))
    inv = torch.argsort(perm)
    res = {}

    def evaluate(dt, permute):
        cfg_d = ref_config("3dmatch", 20, 200.0)
        cfg_d.kpfcn_config["architecture"] = architectures["3dmatch"]
        pipe = Pipeline(cfg_d)
        sd = pipe.state_dict()
        for k, a in Wnp.items():
            sd[k] = T(a)
        pipe.load_state_dict(sd)
        pipe = pipe.to(dt)
        head, tr = pipe.denoising_coarse_matching, pipe.denoising_transformer
        with torch.no_grad():
            pe_s, pe_t = tr.positional_encoding(T(fwd["b1_src_warped"]).to(dt)), tr.positional_encoding(cb["p_t"].to(dt))
            if permute:
                head.src_proj.weight.copy_(head.src_proj.weight[perm].clone())
                assert head.src_proj.bias is None
                pe_s, pe_t = pe_s[:, :, perm].contiguous(), pe_t[:, :, perm].contiguous()
        s = T(up["branch_out_src32"]).to(dt).requires_grad_(True)
        t = T(up["branch_out_tgt32"]).to(dt).requires_grad_(True)
        hat, _ = head(s, t, pe_s, pe_t, cb["src_mask"], cb["tgt_mask"], {}, pe_type="rotary")
        gt = torch.zeros_like(hat)
        gt[0][cb["matches"][0][0], cb["matches"][0][1]] = 1
        loss = crit.compute_correspondence_loss(hat, gt)
        loss.backward()
        gw = head.src_proj.weight.grad
        if permute:
            gw = gw[inv]                                  # row perm[i] of the original weight sits in row i: back to the original order
        return dict(conf=hat.detach().numpy(), loss=np.float64(float(loss)), up_src=s.grad.numpy(), up_tgt=t.grad.numpy(),
                    g_src_proj_weight=gw[::6, ::6].numpy().copy(), g_bin_score=head.bin_score.grad.numpy().copy())

    for tag, dt, pm in (("h64", torch.float64, False), ("h32", torch.float32, False), ("h32perm", torch.float32, True)):
        for k, a in evaluate(dt, pm).items():
            res["%s_%s" % (tag, k)] = a
    # the shipped float32 evaluation IS the end-to-end run's head (same inputs, same code)
    assert np.array_equal(res["h32_up_src"], up["branch_up_src32"]) and np.array_equal(res["h32_up_tgt"], up["branch_up_tgt32"])
    for k in ("conf", "up_src", "up_tgt", "g_src_proj_weight", "g_bin_score"):
        a64 = res["h64_" + k]
        m = max(float(np.abs(a64).max()), 1e-30)
        print("%-18s max %.3e   |h32 - h64| / max %.3e   |h32perm - h64| / max %.3e" % (
            k, m, float(np.abs(res["h32_" + k] - a64).max()) / m, float(np.abs(res["h32perm_" + k] - a64).max()) / m))
    keep = {k: (a.astype(np.float32) if (k.startswith("h64_") and k.endswith("conf")) else a) for k, a in res.items() if not k.startswith("h32_up_")}
    out = os.path.join(GOLD, "train_backward_head_control.npz")
    np.savez_compressed(out, **keep)
    print("wrote", out, os.path.getsize(out), "bytes;", len(keep), "arrays")


if __name__ == "__main__":
    main()
