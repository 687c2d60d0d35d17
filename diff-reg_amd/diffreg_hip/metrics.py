"""Host-side mirror of the reference's evaluation harness on the HIP kernels.

Same names, arguments and return conventions as the functions the reference's testers call after `Pipeline.forward`:

    MatchMotionLoss.compute_inlier_ratio / ransac_regist_coarse / compute_registration_recall   3D/models/loss.py:347-448
    compute_nrfmr                                                                                  3D/lib/tester.py:150-210

so that `from diffreg_hip.metrics import MatchMetrics as MML, compute_nrfmr` is the only edit in `lib/tester.py`
(INTEGRATION.md).  Everything runs on the device through libdiffreg_hip.so (dr_inlier_ratio_f32, dr_nrfmr_f32,
dr_ransac_corr_f64, dr_registration_recall_f64); there is no CPU path and no Open3D / nibabel dependency.
`evaluate_pairs` is the batched form the sharded benchmark uses: P pairs per call, no host synchronisation.
"""
import torch

from . import lib


def segment_matches(match_pred, B):
    """the reference's flat [K,3] (b,i,j) list (rows grouped by b, as nonzero() returns them) -> ([B,K,3], count [B]) without
    a host sync (cap = K)."""
    K = match_pred.shape[0]
    dev = match_pred.device
    # the reference's functions take the rows in any order; ranks inside a pair need them grouped by b: stable sort (a no-op
    # for lists that nonzero() produced)
    order = torch.argsort(match_pred[:, 0], stable=True)
    match_pred = match_pred[order]
    b = match_pred[:, 0]
    count = torch.bincount(b, minlength=B)[:B]
    start = torch.cumsum(count, 0) - count
    rank = torch.arange(K, device=dev) - start[b]
    seg = torch.zeros(B, max(K, 1), 3, dtype=torch.int64, device=dev)
    seg[b, rank] = match_pred
    return seg, count.to(torch.int32)


class MatchMetrics:
    """static metric methods of models.loss.MatchMotionLoss, on device"""

    RANSAC_ITERS = 50000          # RANSACConvergenceCriteria(50000, 1000), loss.py:23
    RANSAC_SEED = 0

    @staticmethod
    def compute_inlier_ratio(match_pred, data, inlier_thr, s2t_flow=None):
        s_pcd, t_pcd = data["s_pcd"], data["t_pcd"]
        seg, count = segment_matches(match_pred, len(s_pcd))
        ir, _ = lib.inlier_ratio(seg, count, s_pcd, t_pcd, data["batched_rot"], data["batched_trn"], inlier_thr, s2t_flow)
        return ir

    @staticmethod
    def ransac_regist_coarse(batched_src_pcd, batched_tgt_pcd, src_mask, tgt_mask, match_pred, seed=None, iters=None,
                             distance_threshold=0.05):
        """-> (rot [B,3,3], trn [B,3,1]) float64.  src_mask / tgt_mask only bound the valid prefix in the reference
        (loss.py:358-359); indices in match_pred already address the padded clouds, so they are not needed here."""
        seg, count = segment_matches(match_pred, len(batched_src_pcd))
        r = lib.ransac_corr(seg, count, batched_src_pcd, batched_tgt_pcd, distance_threshold,
                            MatchMetrics.RANSAC_ITERS if iters is None else iters,
                            MatchMetrics.RANSAC_SEED if seed is None else seed)
        return r["rot"], r["trn"]

    @staticmethod
    def compute_registration_recall(R_est, t_est, data, thr=0.2):
        if data.get("gt_cov") is None:
            return 0.0
        info = torch.as_tensor(data["gt_cov"] if torch.is_tensor(data["gt_cov"]) else
                               torch.stack([torch.as_tensor(g, dtype=torch.float64) for g in data["gt_cov"]]))
        _, ok = lib.registration_recall(R_est, t_est, data["batched_rot"], data["batched_trn"], info.to(R_est.device), thr)
        return float(ok.sum().item()) / len(R_est)


def compute_nrfmr(match_pred, data, recall_thr=0.04):
    """mean over the pairs of the fraction of metric points recalled (tester.py:150-210)"""
    s_pcd, t_pcd = data["s_pcd"], data["t_pcd"]
    raws, flows, idxs = data["src_pcd_list"], data["sflow_list"], data["metric_index_list"]
    dev = s_pcd.device
    B = len(raws)
    seg, count = segment_matches(match_pred, B)
    off = lambda lens: torch.tensor([0] + list(torch.tensor(lens).cumsum(0).tolist()), dtype=torch.int32, device=dev)
    q_len = [len(i) for i in idxs]
    r, _ = lib.nrfmr(seg, count, s_pcd[:B], t_pcd[:B], torch.cat(list(raws)), torch.cat(list(flows)), off([len(x) for x in raws]),
                     torch.cat([i.to(torch.int64) for i in idxs]), off(q_len), max(q_len), data["batched_rot"][:B],
                     data["batched_trn"][:B], 0.1, recall_thr)
    return r.sum() / B


def evaluate_pairs(matches, count, s_pcd, t_pcd, rot_gt, trn_gt, info=None, inlier_thr=0.1, fmr_thr=0.05, ransac_iters=50000,
                   ransac_thr=0.05, rr_thr=0.2, seed=0, pair_ids=None):
    """The 3DMatch tester's per-pair work (3D/lib/tester.py:73-85, 116-118) for P pairs, asynchronously on the current stream:
    IR (thr 0.1), FMR flag (IR > 0.05), correspondence RANSAC, registration recall.  -> dict of device tensors [P, ...]."""
    ir, n_inl = lib.inlier_ratio(matches, count, s_pcd, t_pcd, rot_gt, trn_gt, inlier_thr)
    rs = lib.ransac_corr(matches, count, s_pcd, t_pcd, ransac_thr, ransac_iters, seed, pair_ids)
    out = dict(ir=ir, n_inlier=n_inl, fmr=(ir > fmr_thr).float(), **rs)
    if info is not None:
        out["rr_err"], out["rr_ok"] = lib.registration_recall(rs["rot"], rs["trn"], rot_gt, trn_gt, info, rr_thr)
    return out
