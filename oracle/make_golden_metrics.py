"""Mint golden vectors for the evaluation-harness row (f2) by RUNNING THE REFERENCE's own metric functions
(build container only; needs /root/reference).

    python oracle/make_golden_metrics.py        # writes tests/golden/metrics_ref.npz

Imported from where they lie: MatchMotionLoss.compute_inlier_ratio / compute_registration_recall / computeTransformationErr
(3D/models/loss.py) and compute_nrfmr / blend_anchor_motion (3D/lib/tester.py).  Shims: MagicMock for open3d, tensorboardX,
easydict, cv2, sklearn-free nothing else; `nibabel.quaternions.mat2quat` is not installed, so the reference's
computeTransformationErr runs with oracle.metrics_oracle.mat2quat injected (stated in that module's header).
Inputs come from diffreg_hip.synth (integer hash); only reference OUTPUTS are stored.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "metrics_ref.npz")
TREE = "/root/reference/Diff-Reg-3dmatch"
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)

CASES = [(256, 256, 3), (200, 256, 5), (512, 512, 8)]      # (N, M, seed)


def scene(N, M, seed):
    """The inputs of one case: shared with tests/test_metrics_*.py through this function's twin in tests/helpers.py"""
    from tests.helpers import metrics_scene
    return metrics_scene(N, M, seed)


def main():
    from oracle import metrics_oracle as mo
    for m in ("open3d", "tensorboardX", "easydict", "cv2"):
        sys.modules[m] = MagicMock()
    nib = types.ModuleType("nibabel")
    nibq = types.ModuleType("nibabel.quaternions")
    nibq.mat2quat = mo.mat2quat
    nib.quaternions = nibq
    sys.modules["nibabel"], sys.modules["nibabel.quaternions"] = nib, nibq
    os.chdir(TREE)
    sys.path.insert(0, TREE)
    from models.loss import MatchMotionLoss as MML, computeTransformationErr
    from lib.tester import compute_nrfmr, blend_anchor_motion

    out = {}
    for (N, M, seed) in CASES:
        sc = scene(N, M, seed)
        tag = "%dx%d_s%d_" % (N, M, seed)
        data = dict(s_pcd=sc["s_pcd"], t_pcd=sc["t_pcd"], batched_rot=sc["rot"], batched_trn=sc["trn"])
        out[tag + "ir3d"] = MML.compute_inlier_ratio(sc["matches"], data, inlier_thr=0.1).numpy()
        d4 = dict(data, t_pcd=sc["t_pcd4"], src_pcd_list=[sc["raw_pcd"]], sflow_list=[sc["raw_flow"]],
                  metric_index_list=[sc["metric_index"]])
        out[tag + "ir4d"] = MML.compute_inlier_ratio(sc["matches"], d4, inlier_thr=0.04, s2t_flow=sc["coarse_flow"][None]).numpy()
        out[tag + "nrfmr"] = np.float32(compute_nrfmr(sc["matches"], d4, recall_thr=0.04))
        m = sc["matches"]
        anchors = sc["s_pcd"][0][m[:, 1]]
        motion = sc["t_pcd4"][0][m[:, 2]] - anchors
        bl, valid = blend_anchor_motion(sc["raw_pcd"][sc["metric_index"]].numpy(), anchors.numpy(), motion.numpy(), knn=3,
                                        search_radius=0.1)
        out[tag + "blended"] = bl
        out[tag + "blend_valid"] = valid
        # registration recall: estimates at growing distance from the ground truth
        errs, oks = [], []
        for k, (Re, te) in enumerate(sc["est"]):
            dk = dict(batched_rot=sc["rot"], batched_trn=sc["trn"], gt_cov=[sc["info"]])
            oks.append(MML.compute_registration_recall(Re[None], te[None], dk, thr=0.2))
            gt, pr = np.eye(4), np.eye(4)
            gt[:3, :3], gt[:3, 3:] = sc["rot"][0].numpy(), sc["trn"][0].numpy()
            pr[:3, :3], pr[:3, 3:] = Re.numpy(), te.numpy()
            errs.append(computeTransformationErr(np.linalg.inv(gt) @ pr, sc["info"]))
        out[tag + "rr_err"] = np.array(errs)
        out[tag + "rr_ok"] = np.array(oks)
        print(tag, "ir3d %.4f ir4d %.4f nrfmr %.4f rr %s" % (out[tag + "ir3d"][0], out[tag + "ir4d"][0], out[tag + "nrfmr"], oks))
    # batch_mutual_topk_select of the 2D-3D tree (vision3d/ops/mutual_topk_select.py), loaded from the file where it lies
    import importlib.util
    torch.Tensor.cuda = lambda self, *a, **k: self            # the function hard-codes .cuda()
    spec = importlib.util.spec_from_file_location("ref_mts", "/root/reference/Diff-Reg-2d3d/vision3d/ops/mutual_topk_select.py")
    mts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mts)
    from tests.helpers import topk_case
    for name in ("patch64_k2_thr", "ragged_k3_or", "smallest_k1", "masked_k2"):
        c = topk_case(name)
        b, i, j, s = mts.batch_mutual_topk_select(c["score"], c["k"], row_masks=c["row_masks"], col_masks=c["col_masks"],
                                                  largest=c["largest"], threshold=c["threshold"], mutual=c["mutual"])
        out["mts_" + name + "_idx"] = torch.stack([b, i, j], 1).numpy()
        out["mts_" + name + "_score"] = s.numpy()
        print("mutual_topk", name, len(b))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
