"""ORACLE -- CPU restatement of the patch-correspondence block behind the 2D-3D loop.  TEST INFRASTRUCTURE.

Only tests/ may import this module.  Citations: EXP = /root/reference/Diff-Reg-2d3d/experiments/2d3dmatr.rgbdv2.stage4.level3.stage1,
V3D = /root/reference/Diff-Reg-2d3d/vision3d.  The three helper ops can be replaced by the reference's own functions (`ops=`): that is how
oracle/make_golden_fine2d3d.py mints tests/golden/fine2d3d.npz, and tests/test_fine2d3d_oracle.py pins the restated ops below to it.
"""
import torch


def index_select(inputs, indices, dim):
    """V3D/ops/index_select.py:4-33"""
    out = inputs.index_select(dim, indices.reshape(-1))
    if indices.dim() > 1:
        out = out.view(*(inputs.shape[:dim] + indices.shape + inputs.shape[dim + 1:]))
    return out


def pairwise_cosine_similarity(x, y, normalized=False):
    """V3D/ops/cosine_similarity.py:34-66"""
    if not normalized:
        x = torch.nn.functional.normalize(x, p=2, dim=-1)
        y = torch.nn.functional.normalize(y, p=2, dim=-1)
    return 0.5 * (torch.matmul(x, y.transpose(-1, -2)) + 1.0)


def batch_mutual_topk_select(score_mat, k, row_masks=None, col_masks=None, threshold=None, largest=True, mutual=True):
    """V3D/ops/mutual_topk_select.py:63-134 (reduce_result = True)"""
    B, N, M = score_mat.shape
    bi = torch.arange(B)
    s = score_mat                                              # (the masks are applied to the selection, not to the scores: :117-120)
    ri = s.topk(k=k, dim=2, largest=largest)[1]
    rm = torch.zeros_like(s, dtype=torch.bool)
    rm[bi.view(B, 1, 1).expand(-1, N, k), torch.arange(N).view(1, N, 1).expand(B, -1, k), ri] = True
    ci = s.topk(k=k, dim=1, largest=largest)[1]
    cm = torch.zeros_like(s, dtype=torch.bool)
    cm[bi.view(B, 1, 1).expand(-1, k, M), ci, torch.arange(M).view(1, 1, M).expand(B, k, -1)] = True
    corr = torch.logical_and(rm, cm) if mutual else torch.logical_or(rm, cm)
    if threshold is not None:
        corr = torch.logical_and(corr, torch.gt(s, threshold) if largest else torch.lt(s, threshold))
    if row_masks is not None:
        corr = torch.logical_and(corr, row_masks.unsqueeze(-1))
    if col_masks is not None:
        corr = torch.logical_and(corr, col_masks.unsqueeze(1))
    b, r, c = torch.nonzero(corr, as_tuple=True)
    return b, r, c, score_mat[b, r, c]


OPS = dict(index_select=index_select, pairwise_cosine_similarity=pairwise_cosine_similarity, batch_mutual_topk_select=batch_mutual_topk_select)


def extract_patch_correspondences(img_node_corr_indices, pcd_node_corr_indices, img_node_levels, all_img_total_nodes, all_img_node_knn_indices,
                                  pcd_node_knn_indices, pcd_node_knn_masks, img_feats_f, pcd_feats_f, img_points_f, img_pixels_f, pcd_points_f,
                                  pcd_pixels_f, ops=OPS, trace=None):
    """EXP/model.py:699-774"""
    sel, cos, topk = ops["index_select"], ops["pairwise_cosine_similarity"], ops["batch_mutual_topk_select"]
    img_node_corr_levels = img_node_levels[img_node_corr_indices]                                          # :699
    pcd_padded_feats_f = torch.cat([pcd_feats_f, torch.zeros_like(pcd_feats_f[:1])], dim=0)                 # :707
    all_i, all_p = [], []
    for i in range(len(all_img_node_knn_indices)):                                                          # :714
        m = torch.eq(img_node_corr_levels, i)
        if m.sum().item() == 0:
            continue
        cur_img = img_node_corr_indices[m] - all_img_total_nodes[i]
        cur_pcd = pcd_node_corr_indices[m]
        ik = sel(all_img_node_knn_indices[i], cur_img, dim=0)                                               # :726
        im = torch.ones_like(ik, dtype=torch.bool)                                                          # :727
        fi = sel(img_feats_f, ik, dim=0)                                                                    # :728
        pk, pm = pcd_node_knn_indices[cur_pcd], pcd_node_knn_masks[cur_pcd]                                 # :730-731
        fp = sel(pcd_padded_feats_f, pk, dim=0)                                                             # :732
        sim = cos(fi, fp, normalized=True)                                                                  # :734-736
        b, r, c, _ = topk(sim, k=2, row_masks=im, col_masks=pm, threshold=0.75, largest=True, mutual=True)  # :738-746
        if trace is not None:
            trace.append(dict(level=i, similarity=sim, batch=b, row=r, col=c))
        all_i.append(ik[b, r])
        all_p.append(pk[b, c])
    img_corr = torch.cat(all_i, 0) if all_i else torch.zeros(0, dtype=torch.int64)
    pcd_corr = torch.cat(all_p, 0) if all_p else torch.zeros(0, dtype=torch.int64)
    n_f = pcd_points_f.shape[0]
    uniq = torch.unique(img_corr * n_f + pcd_corr)                                                          # :759-761
    img_corr = torch.div(uniq, n_f, rounding_mode="floor")
    pcd_corr = uniq % n_f
    ip, ix = img_points_f.view(-1, 3), img_pixels_f.view(-1, 2)
    return dict(img_node_corr_levels=img_node_corr_levels, img_corr_indices=img_corr, pcd_corr_indices=pcd_corr, img_corr_points=ip[img_corr],
                img_corr_pixels=ix[img_corr], pcd_corr_points=pcd_points_f[pcd_corr], pcd_corr_pixels=pcd_pixels_f[pcd_corr],
                corr_scores=(img_feats_f[img_corr] * pcd_feats_f[pcd_corr]).sum(1))
