"""oracle/fine2d3d_oracle.py (restated index_select / pairwise_cosine_similarity / batch_mutual_topk_select inside the block of
EXP/model.py:699-774) against the vectors minted with the reference's own ops (tests/golden/fine2d3d.npz).  CPU only."""
import os

import numpy as np
import torch

from oracle import fine2d3d_oracle as fo
from tests.helpers import fine2d3d_case

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fine2d3d.npz"))


def test_block_matches_reference_ops():
    tr = []
    o = fo.extract_patch_correspondences(trace=tr, **fine2d3d_case())
    for t in tr:
        assert np.abs(t["similarity"].numpy() - G["sim_level%d" % t["level"]]).max() < 1e-6
        assert np.array_equal(torch.stack([t["batch"], t["row"], t["col"]], 1).numpy(), G["sel_level%d" % t["level"]])
    for k in ("img_node_corr_levels", "img_corr_indices", "pcd_corr_indices", "img_corr_points", "img_corr_pixels", "pcd_corr_points", "pcd_corr_pixels"):
        assert np.array_equal(o[k].numpy(), G[k]), k
    assert np.abs(o["corr_scores"].numpy() - G["corr_scores"]).max() < 1e-6
    assert len(G["img_corr_indices"]) > 500 and (G["corr_scores"] > 0.5).all()        # the case is not vacuous
