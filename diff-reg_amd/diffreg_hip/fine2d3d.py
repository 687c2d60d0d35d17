"""The patch-correspondence block behind the 2D-3D loop on the device: `EXP/model.py:699-780` of
Diff-Reg-2d3d (EXP = experiments/2d3dmatr.rgbdv2.stage4.level3.stage1), same variable names.  Per image level: the fine
features of the two patches of every node correspondence are gathered and compared (dr_patch_similarity_f32), the mutual
top-k entries beyond the threshold are selected (dr_mutual_topk_select_f32); then duplicates are removed (dr_unique_pairs_i64)
and the correspondences' points, pixels and scores gathered (dr_corr_gather_f32).

    out = extract_patch_correspondences(img_node_corr_indices, pcd_node_corr_indices, img_node_levels, all_img_total_nodes,
                                        all_img_node_knn_indices, pcd_node_knn_indices, pcd_node_knn_masks, img_feats_f, pcd_feats_f,
                                        img_points_f, img_pixels_f, pcd_points_f, pcd_pixels_f)
    output_dict.update(out)           # replaces model.py:699-774
"""
import torch

from . import lib


def extract_patch_correspondences(img_node_corr_indices, pcd_node_corr_indices, img_node_levels, all_img_total_nodes, all_img_node_knn_indices,
                                  pcd_node_knn_indices, pcd_node_knn_masks, img_feats_f, pcd_feats_f, img_points_f, img_pixels_f, pcd_points_f,
                                  pcd_pixels_f, k=2, threshold=0.75, mutual=True):
    dev = img_feats_f.device
    img_node_corr_levels = img_node_levels[img_node_corr_indices]                                          # model.py:699
    out = {"img_node_corr_indices": img_node_corr_indices, "pcd_node_corr_indices": pcd_node_corr_indices,
           "img_node_corr_levels": img_node_corr_levels}
    num_points_f = pcd_points_f.shape[0]
    all_img_corr_indices, all_pcd_corr_indices = [], []
    for i, img_node_knn_indices in enumerate(all_img_node_knn_indices):                                    # :714
        node_corr_masks = torch.eq(img_node_corr_levels, i)
        if node_corr_masks.sum().item() == 0:                                                               # :717 (the reference's host read)
            continue
        cur_img = img_node_corr_indices[node_corr_masks] - all_img_total_nodes[i]                           # :720-721
        cur_pcd = pcd_node_corr_indices[node_corr_masks]
        img_node_corr_knn_indices = img_node_knn_indices[cur_img]                                           # (P, Ki)   :726
        pcd_node_corr_knn_indices = pcd_node_knn_indices[cur_pcd]                                           # (P, Kc)   :730
        pcd_node_corr_knn_masks = pcd_node_knn_masks[cur_pcd]                                               # :731
        # index `num_points_f` is the zero row of pcd_padded_feats_f (:707)
        similarity_mat = lib.patch_similarity(img_feats_f, img_node_corr_knn_indices, pcd_feats_f, pcd_node_corr_knn_indices)   # :728-738
        batch_indices, row_indices, col_indices, _ = lib.batch_mutual_topk_select(                          # :740-748
            similarity_mat, k=k, row_masks=None, col_masks=pcd_node_corr_knn_masks, threshold=threshold, largest=True, mutual=mutual)
        all_img_corr_indices.append(img_node_corr_knn_indices[batch_indices, row_indices])                  # :750-751
        all_pcd_corr_indices.append(pcd_node_corr_knn_indices[batch_indices, col_indices])
    if all_img_corr_indices:
        img_corr_indices = torch.cat(all_img_corr_indices, dim=0)
        pcd_corr_indices = torch.cat(all_pcd_corr_indices, dim=0)
    else:
        img_corr_indices = pcd_corr_indices = torch.zeros(0, dtype=torch.int64, device=dev)
    keys, count = lib.unique_pairs(img_corr_indices, pcd_corr_indices, num_points_f)                        # :759-763
    out.update(lib.corr_gather(keys, count, num_points_f, img_points_f, img_pixels_f, pcd_points_f, pcd_pixels_f, img_feats_f, pcd_feats_f))   # :765-774
    return out


def registration_with_pnp_ransac(corr_points, corr_pixels, intrinsics, distortion=None, num_iterations=5000, distance_tolerance=8.0, transposed=True,
                                 seed=0):
    """vision3d.utils.opencv.registration_with_pnp_ransac (opencv.py:10-63; called by EXP/eval.py:174-182 with 50000 iterations, tolerance 8) on the
    device: same arguments, returns the estimated 4 x 4 transform (3D -> camera) as a numpy array, or None with fewer than 4 correspondences.
    Inputs may be numpy arrays (as in the reference) or device tensors.  No lens distortion (the reference passes none)."""
    import numpy as np
    if distortion is not None and np.any(np.asarray(distortion) != 0):
        raise NotImplementedError("lens distortion")
    dev = corr_points.device if torch.is_tensor(corr_points) and corr_points.is_cuda else torch.device("cuda", torch.cuda.current_device())
    P = torch.as_tensor(np.asarray(corr_points) if not torch.is_tensor(corr_points) else corr_points).to(dev, torch.float32)
    px = torch.as_tensor(np.asarray(corr_pixels) if not torch.is_tensor(corr_pixels) else corr_pixels).to(dev, torch.float32)
    Kc = intrinsics.detach().cpu().numpy() if torch.is_tensor(intrinsics) else np.asarray(intrinsics)
    r = lib.pnp_ransac(P, px, Kc, num_iterations, distance_tolerance, seed, transposed)
    return None if r is None else r["transform"].cpu().numpy()
