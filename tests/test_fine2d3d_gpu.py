"""The patch-correspondence block behind the 2D-3D loop on the device (row f4: dr_patch_similarity_f32, dr_mutual_topk_select_f32,
dr_unique_pairs_i64, dr_corr_gather_f32 through diffreg_hip.fine2d3d) against vectors minted with the reference's own ops.  Needs a GPU."""
import os

import numpy as np
import pytest
import torch

from diffreg_hip import lib
from diffreg_hip.fine2d3d import extract_patch_correspondences
from tests.helpers import fine2d3d_case, guarded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fine2d3d.npz"))


def to_dev(c):
    return {k: ([t.to(DEV) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(DEV) if torch.is_tensor(v) else v)) for k, v in c.items()}


def test_patch_similarity_against_reference():
    c = fine2d3d_case()
    lev = c["img_node_levels"][c["img_node_corr_indices"]]
    fi, fp = c["img_feats_f"].to(DEV), c["pcd_feats_f"].to(DEV)
    for li, knn in enumerate(c["all_img_node_knn_indices"]):
        m = lev == li
        ik = knn[c["img_node_corr_indices"][m] - c["all_img_total_nodes"][li]]
        pk = c["pcd_node_knn_indices"][c["pcd_node_corr_indices"][m]]
        P, Ki = ik.shape
        out, chk = guarded((P, Ki, pk.shape[1]), torch.float32, DEV, fill=float("nan"))
        ikd, pkd = ik.to(DEV), pk.to(DEV)
        lib.check(lib.raw().dr_patch_similarity_f32(P, Ki, pk.shape[1], 128, lib.ptr(fi), lib.ptr(ikd), lib.ptr(fp), lib.ptr(pkd), fp.shape[0], lib.ptr(out), None))
        torch.cuda.synchronize()
        chk()
        assert np.abs(out.cpu().numpy() - G["sim_level%d" % li]).max() < 2e-6


def test_block_against_reference():
    c = fine2d3d_case()
    o = extract_patch_correspondences(**to_dev(c))
    # the selection thresholds similarities at 0.75: entries within 1e-5 of it (or of a k-th best) could flip -- none in this case
    for k in ("img_node_corr_levels", "img_corr_indices", "pcd_corr_indices", "img_corr_points", "img_corr_pixels", "pcd_corr_points", "pcd_corr_pixels"):
        assert np.array_equal(o[k].cpu().numpy(), G[k]), k
    assert np.abs(o["corr_scores"].cpu().numpy() - G["corr_scores"]).max() < 2e-6


@pytest.mark.parametrize("n,span", [(1, 5), (1000, 50), (70000, 3000), (5, 1)])
def test_unique_pairs_against_torch_unique(n, span):
    g = torch.Generator().manual_seed(n)
    a = torch.randint(0, span, (n,), generator=g)
    b = torch.randint(0, span, (n,), generator=g)
    keys, cnt = lib.unique_pairs(a.to(DEV), b.to(DEV), 100003)
    want = torch.unique(a * 100003 + b)
    assert int(cnt) == want.numel() and torch.equal(keys[: int(cnt)].cpu(), want)


def test_empty_and_unsupported():
    z = torch.zeros(0, dtype=torch.int64, device=DEV)
    keys, cnt = lib.unique_pairs(z, z, 7)
    assert int(cnt) == 0
    f = torch.zeros(4, 8, device=DEV)
    with pytest.raises(RuntimeError):
        lib.patch_similarity(f, torch.zeros(1, 4, dtype=torch.int64, device=DEV), f, torch.zeros(1, 200, dtype=torch.int64, device=DEV))    # Kc > 128
