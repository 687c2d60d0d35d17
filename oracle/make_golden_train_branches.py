"""Mint the gradients of the training graphs for the configuration branches NO shipped yaml selects, by RUNNING THE REFERENCE in float32 and in
float64 (build container only; needs /root/reference):

    python oracle/make_golden_train_branches.py       # writes tests/golden/train_backward_branches.npz

Forms (3D/models/transformero.py:50-57, 234-254; matching.py:181, 193-205; position_encoding.py:36-44):
    sin      pe_type 'sinusoidal', disentangled   q = W_q (x + pe), k = W_k (y + pe), v = W_v y; the head adds the code behind src_proj
    rot_ent  pe_type 'rotary',     entangled      the features are rotated ONCE in front of the layers, the layers and the head see no code
    sin_ent  pe_type 'sinusoidal', entangled      the code is added once in front of the layers
    dsm      rotary, disentangled, match_type 'dual_softmax'   softmax over rows x softmax over columns of sim / temperature (no bin score)
Per form, on one small synthetic pair (tests/helpers.train_branch_case: 40 x 32 points, soft head; the denoising branch on the source warped by the
ground-truth pose), at a feature scale chosen so that NO ReLU unit of either graph sits within 5e-6 of its kink (see relu_margin below):
    the denoising branch (pipeline.py:209-212 + compute_correspondence_loss): conf, loss, gradients of the backbone features (entries [0, ::3, ::4]), every parameter gradient
    the coarse branch with the motion term (pipeline.py:184-196 + loss.py:97-128; the positioning layer fits inside the transformer for the
    disentangled forms and is skipped by the entangled ones, transformero.py:252): the same
each in float32 (*32*, the reference as shipped) and float64 (*64*, the yardstick: module.double(), with the one dtype-preserving shim of
make_golden_train_grads.py in procrustes.py:41).  Parameter gradients are stored as their [::12, ::12] entries (all entries of 1-D / 0-D tensors); everything as float32 (the float64 run is
the yardstick of bounds at the 1e-3 scale: its rounding to float32 does not matter).
With dual_softmax the loss takes its 'dual_softmax' form (loss.py:273-314) and the Pipeline-level noising path does not exist (pipeline.py:299 reads
bin_score): the two branch graphs are built from the modules, as for every other form.  Only reference OUTPUTS are stored."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLD, "train_backward_branches.npz")
TREE = "/root/reference/Diff-Reg-3dmatch"
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)

TRAIN_FORMS = {"sin": ("sinusoidal", False, "sinkhorn"), "rot_ent": ("rotary", True, "sinkhorn"), "sin_ent": ("sinusoidal", True, "sinkhorn"),
               "dsm": ("rotary", False, "dual_softmax")}


STRIDE = 12                    # parameter gradients are stored as their [::12, ::12] entries (all entries of 1-D / 0-D tensors), float32
CASE = (40, 32, 70)            # N, M, seed of the case (tests/helpers.train_branch_case): small, so that a point off every ReLU kink exists nearby
SCALES = tuple(0.5 + 0.01 * ((k + 1) // 2) * (1 if k % 2 else -1) for k in range(41))      # candidate feature scales: 0.5, 0.51, 0.49, 0.52, ...
MARGIN = 5e-6                  # the smallest |pre-activation| over all ReLU units of both graphs (float64 run) a case must have
DSM_TEMPERATURE = 3.0          # dsmax_temperature of the 'dsm' form: the yaml's 0.1 saturates the synthetic head (conf = 1.0000, loss = 0, gradients ~1e-30)


def sub(g):
    return (g[::STRIDE, ::STRIDE] if g.dim() == 2 else g).detach().numpy().astype(np.float32)


def main():
    import torch
    from oracle.make_golden import ref_config, HEAD_GAIN_SOFT
    from oracle.make_golden_branches import form_config
    from oracle.make_golden_train import LOSS_CFG
    sys.modules["open3d"] = MagicMock()
    for m in ("easydict", "tensorboardX", "nibabel", "nibabel.quaternions", "cv2"):
        sys.modules.setdefault(m, MagicMock())
    torch.Tensor.cuda = lambda self, *a, **k: self
    os.chdir(TREE)
    sys.path.insert(0, TREE)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from diffreg_hip import synth
    from models.pipeline import Pipeline
    from models.loss import MatchMotionLoss
    from models.procrustes import SoftProcrustesLayer
    from configs.models import architectures
    from tests.helpers import train_branch_case

    C = synth.VARIANTS["3dmatch"]["C"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    Wnp = dict(synth.make_weights(C, seed=7, head_gain=HEAD_GAIN_SOFT))
    Wnp.update(synth.make_weights_coarse(C, seed=17, head_gain=HEAD_GAIN_SOFT))
    cb = train_branch_case(*CASE)
    ref_bwp = SoftProcrustesLayer.batch_weighted_procrustes

    def bwp_dtype_preserving(X, Y, w, eps=0.0001):                 # (procrustes.py:17-44 with the cast of :41 following the input dtype: see make_golden_train_grads.py)
        bsize = X.shape[0]
        W1 = torch.abs(w).sum(dim=1, keepdim=True)
        w_norm = w / (W1 + eps)
        mean_X = (w_norm * X).sum(dim=1, keepdim=True)
        mean_Y = (w_norm * Y).sum(dim=1, keepdim=True)
        Sxy = torch.matmul((Y - mean_Y).transpose(1, 2), w_norm * (X - mean_X)).cpu().double()
        U, D, V = Sxy.svd()
        condition = D.max(dim=1)[0] / D.min(dim=1)[0]
        S = torch.eye(3)[None].repeat(bsize, 1, 1).double()
        S[:, 2:3, 2:3] = (U.det() * V.det()).view(-1, 1, 1)
        R = torch.matmul(U, torch.matmul(S, V.transpose(1, 2))).to(X.dtype)
        t = mean_Y.transpose(1, 2) - torch.matmul(R, mean_X.transpose(1, 2))
        return R, t, condition

    def build_pipe(form, dt):
        pe_type, ent, mtype = TRAIN_FORMS[form]
        cfg = form_config(ref_config, pe_type, ent, steps=20, max_cond=200.0, match_type=mtype)
        if mtype == "dual_softmax":
            cfg.coarse_matching["dsmax_temperature"] = cfg.coarse_transformer["feature_matching"]["dsmax_temperature"] = DSM_TEMPERATURE
        cfg.kpfcn_config["architecture"] = architectures["3dmatch"]
        pipe = Pipeline(cfg)
        sd = pipe.state_dict()
        for k, a in Wnp.items():
            if k in sd:                                          # (dual_softmax: no bin_score parameters)
                sd[k] = T(a)
        pipe.load_state_dict(sd)
        return pipe.to(dt)

    def relu_margin(form, scale):
        """the smallest |pre-activation| over every ReLU unit of both branch graphs (float64 and float32 runs), and whether any unit has different
        signs in the two runs.  A test point must not sit ON a kink: the gradient of everything upstream jumps by a finite amount when one unit
        flips (measured: a unit 5.9e-8 from zero in layer 1 moved the layer-0 / layer-1 gradients by 2e-3 of their maxima under a 1e-6 change of
        the position code), and no float32 implementation can be asked to land on the reference's side of it."""
        pre = {}
        for dt in (torch.float32, torch.float64):
            SoftProcrustesLayer.batch_weighted_procrustes = staticmethod(ref_bwp if dt == torch.float32 else bwp_dtype_preserving)
            pipe = build_pipe(form, dt)
            acts = []
            hooks = [lay.mlp[0].register_forward_hook(lambda m, i, o: acts.append(o.detach().double().clone()))
                     for tr in (pipe.denoising_transformer, pipe.coarse_transformer) for lay in tr.layers if hasattr(lay, "mlp")]
            with torch.no_grad():
                fs, ft = (cb["f_s"] * scale).to(dt), (cb["f_t"] * scale).to(dt)
                pipe.denoising_transformer(fs, ft, cb["warped"].to(dt), cb["p_t"].to(dt), cb["src_mask"], cb["tgt_mask"], {})
                pipe.coarse_transformer(fs, ft, cb["p_s"].to(dt), cb["p_t"].to(dt), cb["src_mask"], cb["tgt_mask"], {})
            for h in hooks:
                h.remove()
            pre[dt] = acts
        margin = min(float(a.abs().min()) for a in pre[torch.float64])
        flips = sum(int(((a > 0) != (b > 0)).sum()) for a, b in zip(pre[torch.float32], pre[torch.float64]))
        return margin, flips

    res = {}
    for form, (pe_type, ent, mtype) in TRAIN_FORMS.items():
        # the feature scale of the case: 0.5 as for the shipped-form vectors unless a ReLU unit sits within MARGIN of its kink there
        scale = None
        for cand in SCALES:
            margin, flips = relu_margin(form, cand)
            print(form, "feature scale %.3f: smallest |ReLU pre-activation| %.2e, float32 / float64 sign differences %d" % (cand, margin, flips))
            if margin >= MARGIN and flips == 0:
                scale = cand
                break
        assert scale is not None, form
        res[form + "_feat_scale"], res[form + "_relu_margin"] = np.float64(scale), np.float64(margin)
        for dt, tag in ((torch.float32, "32"), (torch.float64, "64")):
            SoftProcrustesLayer.batch_weighted_procrustes = staticmethod(ref_bwp if dt == torch.float32 else bwp_dtype_preserving)
            pipe = build_pipe(form, dt)
            loss_cfg = dict(LOSS_CFG, match_type=mtype)
            crit = MatchMotionLoss(loss_cfg)
            pre = "%s_" % form
            # ---- denoising branch
            fs_d = (cb["f_s"] * scale).to(dt).clone().requires_grad_(True)
            ft_d = (cb["f_t"] * scale).to(dt).clone().requires_grad_(True)
            warped = cb["warped"].to(dt)
            p_t, p_s = cb["p_t"].to(dt), cb["p_s"].to(dt)
            with torch.enable_grad():
                s_n, t_n, pe_s, pe_t = pipe.denoising_transformer(fs_d, ft_d, warped, p_t, cb["src_mask"], cb["tgt_mask"], {})
                hat, _ = pipe.denoising_coarse_matching(s_n, t_n, pe_s, pe_t, cb["src_mask"], cb["tgt_mask"], {}, pe_type=pe_type)
                gt_d = torch.zeros_like(hat)
                gt_d[0][cb["matches"][0][0], cb["matches"][0][1]] = 1
                loss_d = crit.compute_correspondence_loss(hat, gt_d)
                loss_d.backward()
            f32 = lambda t_: t_.detach().numpy().astype(np.float32)
            res[pre + "branch_conf" + tag], res[pre + "branch_loss" + tag] = f32(hat), np.float64(float(loss_d))
            res[pre + "branch_grad_src" + tag], res[pre + "branch_grad_tgt" + tag] = f32(fs_d.grad)[0, ::3, ::4], f32(ft_d.grad)[0, ::3, ::4]
            n_b = 0
            for k, prm in list(pipe.denoising_transformer.named_parameters()) + [("head." + k2, p2) for k2, p2 in pipe.denoising_coarse_matching.named_parameters()]:
                if prm.grad is not None:
                    res["%sbranch_g%s_%s" % (pre, tag, k)] = sub(prm.grad)
                    n_b += 1
            # ---- coarse branch with the motion term
            for prm in pipe.parameters():
                prm.grad = None
            fs_c = (cb["f_s"] * scale).to(dt).clone().requires_grad_(True)
            ft_c = (cb["f_t"] * scale).to(dt).clone().requires_grad_(True)
            ov = torch.zeros(1, cb["N"], dtype=torch.bool)
            ov[0][cb["matches"][0][0]] = True
            R_gt, t_gt = cb["R_gt"].to(dt), cb["t_gt"].to(dt)
            with torch.enable_grad():
                a_s, a_t, pe_s2, pe_t2 = pipe.coarse_transformer(fs_c, ft_c, p_s, p_t, cb["src_mask"], cb["tgt_mask"], {})
                conf_c, _ = pipe.coarse_matching(a_s, a_t, pe_s2, pe_t2, cb["src_mask"], cb["tgt_mask"], {}, pe_type=pe_type)
                R_c, t_c, _, _, _, _ = pipe.soft_procrustes(conf_c, p_s, p_t, cb["src_mask"], cb["tgt_mask"])
                focal_c = crit.compute_correspondence_loss(conf_c, gt_d)
                wp = (torch.matmul(R_c, p_s.transpose(1, 2)) + t_c).transpose(1, 2)
                wg = (torch.matmul(R_gt, p_s.transpose(1, 2)) + t_gt).transpose(1, 2)
                l1_c = torch.sum(torch.abs((wp - p_s) - (wg - p_s)), 2)[ov].mean()
                (focal_c + 0.1 * l1_c).backward()
            res[pre + "coarse_conf" + tag] = f32(conf_c)
            res[pre + "coarse_R" + tag], res[pre + "coarse_t" + tag] = f32(R_c), f32(t_c)
            res[pre + "coarse_loss" + tag] = np.float64(float(focal_c + 0.1 * l1_c))
            res[pre + "coarse_grad_src" + tag], res[pre + "coarse_grad_tgt" + tag] = f32(fs_c.grad)[0, ::3, ::4], f32(ft_c.grad)[0, ::3, ::4]
            n_c = 0
            for k, prm in list(pipe.coarse_transformer.named_parameters()) + [("head." + k2, p2) for k2, p2 in pipe.coarse_matching.named_parameters()]:
                if prm.grad is not None:
                    res["%scoarse_g%s_%s" % (pre, tag, k)] = sub(prm.grad)
                    n_c += 1
            print(form, "dtype", tag, "branch params", n_b, "coarse params", n_c, "loss_d %.6f" % float(loss_d), "conf max %.4f" % float(hat.max()),
                  "coarse loss %.6f" % float(focal_c + 0.1 * l1_c))
    SoftProcrustesLayer.batch_weighted_procrustes = staticmethod(ref_bwp)
    worst = 0.0
    for k in [k for k in res if "_g32_" in k]:
        a, b = res[k].astype(np.float64), res[k.replace("_g32_", "_g64_")]
        worst = max(worst, float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
    print("largest |g32 - g64| / max|g64| over all parameter tensors: %.3e" % worst)
    res["dsm_temperature"] = np.float64(DSM_TEMPERATURE)
    res["stride"] = np.int64(STRIDE)
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(res), "arrays")


if __name__ == "__main__":
    main()
