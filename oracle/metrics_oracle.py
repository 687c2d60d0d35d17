"""CPU restatement of the evaluation harness that consumes the loop's match_pred.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline legs of the bench tools, never by
the product path (diff-reg_amd/).  Citations: 3D/ = /root/reference/Diff-Reg-3dmatch/.

Pinning (oracle/make_golden_metrics.py -> tests/golden/metrics_ref.npz, checked by tests/test_metrics_oracle.py):
  * inlier_ratio, blend_anchor_motion / nrfmr, transformation_err / registration_recall are held to the reference's own
    functions run in the build container;
  * mat2quat restates nibabel.quaternions.mat2quat (nibabel is a dependency of the reference that is not installed here and
    not vendored; 3D/eccv24_3d_env.yml pins it) -- the reference's computeTransformationErr is run with THIS mat2quat injected,
    so the quaternion step itself is checked only against rotation-matrix identities: **parity unpinned** for that step;
  * ransac_corr restates Open3D 0.13.0 (3D/eccv24_3d_env.yml:139) registration_ransac_based_on_correspondence as published:
    per iteration draw ransac_n correspondences with replacement, TransformationEstimationPointToPoint(with_scaling=False)
    = Eigen::umeyama, score = number of correspondences closer than the threshold, keep the result with the higher fitness,
    ties by the lower inlier RMSE; RANSACConvergenceCriteria(50000, 1000) clamps the confidence to 1.0 so no early exit.
    Open3D is absent and its generator is unseeded (the reference repeats its evaluation 3 times for that reason,
    3D/lib/tester.py:24): **parity unpinned**, anchored on the call site 3D/models/loss.py:13-24 and on properties
    (recovery of the generating pose, invariance to the order of the correspondences, agreement of fitness with a recount).
    Deliberate deviation, shared with the HIP kernel: triples that repeat a source or a target point are skipped (their
    covariance has rank 1, the fitted roll angle would be whatever the SVD routine returns).
"""
import numpy as np
import torch

from diffreg_hip import synth


# ------------------------------------------------------------------------------------------
# inlier ratio  (3D/models/loss.py:383-410)
# ------------------------------------------------------------------------------------------
def inlier_ratio(match_pred, s_pcd, t_pcd, rot, trn, inlier_thr, s2t_flow=None):
    """match_pred [K,3] int64 (b,i,j); s_pcd [B,N,3], t_pcd [B,M,3], rot [B,3,3], trn [B,3,1] float32 -> IR [B] float32"""
    src = s_pcd if s2t_flow is None else s_pcd + s2t_flow
    moved = (torch.matmul(rot, src.transpose(1, 2)) + trn).transpose(1, 2)
    a = moved[match_pred[:, 0], match_pred[:, 1]]
    b = t_pcd[match_pred[:, 0], match_pred[:, 2]]
    hit = torch.sum((a - b) ** 2, dim=1) < inlier_thr ** 2
    out = []
    for k in range(len(s_pcd)):
        sel = match_pred[:, 0] == k
        n = sel.sum()
        out.append(n.float() * 0 if n < 3 else hit[sel].sum().float() / n)
    return torch.stack(out, dim=0)


# ------------------------------------------------------------------------------------------
# NR-FMR  (3D/lib/tester.py:127-210, 3D/datasets/utils.py:5-40)
# ------------------------------------------------------------------------------------------
def knn_point(k, reference_pts, query_pts, stable=False):
    """k nearest reference points of every query: (distances [Q,k] ascending, indices [Q,k]).
    Equal distances (two matches that share a source point are two anchors at the same place) are ordered however
    np.argpartition leaves them in the reference; stable=True states the HIP kernel's rule instead: lowest match row first."""
    d = np.sum((reference_pts[None, :, :] - query_pts[:, None, :]) ** 2, -1)
    if stable:
        idx = np.argsort(d, axis=1, kind="stable")[:, :k]
        return np.sqrt(np.take_along_axis(d, idx, axis=1)), idx
    part = np.argpartition(d, k, axis=1)[:, :k]
    rows = np.arange(len(query_pts))[:, None]
    idx = part[rows, np.argsort(d[rows, part], axis=1)]
    return np.sqrt(np.take_along_axis(d, idx, axis=1)), idx


def blend_anchor_motion(query_loc, reference_loc, reference_flow, knn=3, search_radius=0.1, stable=False):
    dist, idx = knn_point(knn, reference_loc, query_loc, stable)
    dist[dist < 1e-10] = 1e-10
    far = dist > search_radius
    dist[far] = 1e+10
    w = 1.0 / dist
    w = w / np.sum(w, -1, keepdims=True)
    return np.sum(reference_flow[idx] * w.reshape([-1, knn, 1]), axis=1), far.sum(axis=1) < 3


def nrfmr(match_pred, s_pcd, t_pcd, raw_list, flow_list, metric_index_list, rot, trn, recall_thr=0.04, stable=False):
    """-> (mean over pairs, per-pair recall list, per-pair blended motion list); all torch float32 like the reference"""
    per, blends = [], []
    for k in range(len(raw_list)):
        pts = raw_list[k][metric_index_list[k]]
        gt = (torch.matmul(rot[k], (pts + flow_list[k][metric_index_list[k]]).T) + trn[k]).T
        m = match_pred[match_pred[:, 0] == k]
        anchors = s_pcd[k][m[:, 1]]
        motion = t_pcd[k][m[:, 2]] - anchors
        bl, _ = blend_anchor_motion(pts.numpy(), anchors.numpy(), motion.numpy(), knn=3, search_radius=0.1, stable=stable)
        pred = pts + torch.from_numpy(bl).to(pts)
        dist = torch.sqrt(torch.sum((pred - gt) ** 2, dim=1))
        per.append((dist < recall_thr).float().sum() / len(dist))
        blends.append(bl)
    return sum(per) / len(raw_list), per, blends


# ------------------------------------------------------------------------------------------
# correspondence RANSAC  (3D/models/loss.py:13-24, 347-379; Open3D 0.13.0)
# ------------------------------------------------------------------------------------------
def rigid_fit(X, Y):
    """Eigen::umeyama without scaling, batched: X, Y [n,k,3] float64 -> R [n,3,3], t [n,3] with Y ~ R X + t"""
    mx, my = X.mean(1), Y.mean(1)
    sigma = np.einsum("nka,nkb->nab", Y - my[:, None], X - mx[:, None]) / X.shape[1]
    U, _, Vt = np.linalg.svd(sigma)
    d = np.where(np.linalg.det(U) * np.linalg.det(Vt) < 0, -1.0, 1.0)
    U = U.copy()
    U[:, :, 2] *= d[:, None]
    R = U @ Vt
    return R, my - np.einsum("nab,nb->na", R, mx)


def ransac_corr(s_pcd, t_pcd, corr, distance_thr=0.05, iters=50000, seed=0, pair_id=0, chunk=2000):
    """s_pcd [N,3], t_pcd [M,3] float32; corr [K,2] int (src index, tgt index).  Hypothesis h draws the correspondences
    hash_bits(seed, pair_id)[3h + slot] % K.  -> dict(R [3,3], t [3] float64, fitness, inlier_rmse, best_iter, n_inlier)"""
    K = len(corr)
    ident = dict(R=np.eye(3), t=np.zeros(3), fitness=0.0, inlier_rmse=0.0, best_iter=-1, n_inlier=0)
    if K < 3:
        return ident
    S = s_pcd.astype(np.float64)[corr[:, 0]]
    Y = t_pcd.astype(np.float64)[corr[:, 1]]
    draw = (synth.hash_bits(seed, pair_id, iters * 3) % np.uint64(K)).astype(np.int64).reshape(iters, 3)
    si, tj = corr[draw, 0], corr[draw, 1]
    ok = np.ones(iters, bool)
    for a, b in ((0, 1), (0, 2), (1, 2)):
        ok &= (si[:, a] != si[:, b]) & (tj[:, a] != tj[:, b])
    best = (0, 0.0, -1)
    thr2 = distance_thr * distance_thr
    for h0 in range(0, iters, chunk):
        hs = np.nonzero(ok[h0:h0 + chunk])[0] + h0
        if len(hs) == 0:
            continue
        R, t = rigid_fit(S[draw[hs]], Y[draw[hs]])
        d2 = ((np.einsum("nab,kb->nka", R, S) + t[:, None] - Y[None]) ** 2).sum(-1)
        inl = d2 < thr2
        cnt = inl.sum(1)
        err = (d2 * inl).sum(1)
        for n in np.nonzero(cnt >= max(best[0], 1))[0]:
            c, e = int(cnt[n]), float(err[n])
            if c > best[0] or (c == best[0] and e < best[1]):
                best = (c, e, int(hs[n]))
    if best[2] < 0:
        return ident
    R, t = rigid_fit(S[draw[best[2]]][None], Y[draw[best[2]]][None])
    return dict(R=R[0], t=t[0], fitness=best[0] / K, inlier_rmse=float(np.sqrt(best[1] / best[0])), best_iter=best[2],
                n_inlier=best[0])


# ------------------------------------------------------------------------------------------
# registration recall  (3D/models/loss.py:27-44, 415-448)
# ------------------------------------------------------------------------------------------
def mat2quat(Mx):
    """nibabel.quaternions.mat2quat: (w, x, y, z) = the eigenvector of the largest eigenvalue of the symmetric 4x4
    matrix built from the rotation (Bar-Itzhack 2000), sign chosen so that w >= 0."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(Mx, dtype=np.float64).flat
    Kq = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                   [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                   [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                   [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(Kq)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return q * -1 if q[0] < 0 else q


def transformation_err(trans, info):
    er = np.concatenate([trans[:3, 3], mat2quat(trans[:3, :3])[1:]], axis=0)
    return (er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]).item()


def registration_recall(R_est, t_est, rot_gt, trn_gt, infos, thr=0.2):
    """R_est [B,3,3], t_est [B,3,1] (any float), rot_gt/trn_gt float32, infos [B,6,6] -> (recall, errors [B])"""
    errs = []
    for k in range(len(R_est)):
        gt, pr = np.eye(4), np.eye(4)
        gt[:3, :3], gt[:3, 3:] = np.asarray(rot_gt[k]), np.asarray(trn_gt[k]).reshape(3, 1)
        pr[:3, :3], pr[:3, 3:] = np.asarray(R_est[k]), np.asarray(t_est[k]).reshape(3, 1)
        errs.append(transformation_err(np.linalg.inv(gt) @ pr, np.asarray(infos[k])))
    errs = np.array(errs)
    return float((errs <= thr ** 2).sum()) / len(errs), errs


# ------------------------------------------------------------------------------------------
# batch_mutual_topk_select  (Diff-Reg-2d3d/vision3d/ops/mutual_topk_select.py:63-134; the fine matching behind the 2D-3D loop,
# EXP/model.py:744-752).  Pinned by oracle/make_golden_metrics.py (the reference function run on hash-generated scores).
# ------------------------------------------------------------------------------------------
def batch_mutual_topk_select(score_mat, k, row_masks=None, col_masks=None, largest=True, threshold=None, mutual=True):
    """score_mat [B,N,M] -> (batch_indices, row_indices, col_indices, scores), torch.nonzero order"""
    B, N, M = score_mat.shape
    rows = torch.zeros_like(score_mat, dtype=torch.bool)
    rows.scatter_(2, score_mat.topk(k=k, largest=largest, dim=2)[1], True)
    cols = torch.zeros_like(score_mat, dtype=torch.bool)
    cols.scatter_(1, score_mat.topk(k=k, largest=largest, dim=1)[1], True)
    corr = (rows & cols) if mutual else (rows | cols)
    if threshold is not None:
        corr = corr & (score_mat > threshold if largest else score_mat < threshold)
    if row_masks is not None:
        corr = corr & row_masks[:, :, None]
    if col_masks is not None:
        corr = corr & col_masks[:, None, :]
    b, i, j = torch.nonzero(corr, as_tuple=True)
    return b, i, j, score_mat[b, i, j]
