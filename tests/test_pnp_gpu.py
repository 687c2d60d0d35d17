"""dr_pnp_ransac_f64 (row f4: PnP-RANSAC of the 2D-3D fine correspondences) against oracle/pnp_oracle.py on the same hypothesis stream, and as a
drop-in of vision3d.utils.opencv.registration_with_pnp_ransac.  Parity against OpenCV itself is unpinned (cv2 is not in the image).  Needs a GPU."""
import numpy as np
import pytest
import torch

from diffreg_hip import lib
from diffreg_hip.fine2d3d import registration_with_pnp_ransac
from oracle import pnp_oracle as po
from tests.helpers import pnp_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("seed,n,iters", [(1, 400, 512), (2, 400, 1000), (5, 1300, 300), (7, 37, 256)])
def test_same_model_as_the_oracle(seed, n, iters):
    X, px, K, T, good = pnp_scene(seed, n=n)
    want = po.pnp_ransac(X, px, K, num_iterations=iters, distance_tolerance=8.0, seed=seed)
    got = lib.pnp_ransac(torch.from_numpy(X).to(DEV), torch.from_numpy(px).to(DEV), K, iters, 8.0, seed)
    assert int(got["n_inlier"]) == want["n_inlier"] and int(got["best_iter"]) == want["best_iter"]
    assert np.abs(got["transform"].cpu().numpy() - want["transform"]).max() < 1e-6


def test_drop_in_at_the_reference_settings():
    """50 000 hypotheses, tolerance 8 px (EXP/config.py:60-61): the pose of a scene with 40 % outliers; fewer than 4 correspondences -> None"""
    X, px, K, T, good = pnp_scene(11, n=800)
    est = registration_with_pnp_ransac(X, px, K, num_iterations=50000, distance_tolerance=8.0)
    assert est.shape == (4, 4) and np.abs(est[:3, :3] - T[:3, :3]).max() < 2e-3 and np.abs(est[:3, 3] - T[:3, 3]).max() < 6e-3
    assert registration_with_pnp_ransac(X[:3], px[:3], K) is None
    # (w, h) pixel order
    est2 = registration_with_pnp_ransac(torch.from_numpy(X).to(DEV), torch.from_numpy(px[:, ::-1].copy()).to(DEV), K, num_iterations=2000, transposed=False)
    assert np.abs(est2[:3, 3] - T[:3, 3]).max() < 1e-2


def test_minimal_and_degenerate_inputs():
    """fewer than 4 correspondences: no model (the wrapper returns None, as cv2.solvePnPRansac fails); exactly 4, and 20 collinear points (every
    P3P sample degenerate): a finite transform and an inlier count, no launch failure"""
    K = [[500., 0, 320], [0, 500., 240], [0, 0, 1]]
    g = torch.Generator().manual_seed(1)
    for n in (0, 3):
        assert lib.pnp_ransac(torch.zeros(n, 3, device=DEV), torch.zeros(n, 2, device=DEV), K, num_iterations=64) is None
    pts = (torch.randn(4, 3, generator=g) + torch.tensor([0, 0, 5.0])).to(DEV)
    pix = (torch.rand(4, 2, generator=g) * 400).to(DEV)
    r = lib.pnp_ransac(pts, pix, K, num_iterations=256)
    assert torch.isfinite(r["transform"]).all() and 0 <= int(r["n_inlier"]) <= 4
    line = torch.stack([torch.linspace(-1, 1, 20), torch.zeros(20), torch.full((20,), 5.0)], 1).to(DEV)
    lpix = torch.stack([torch.linspace(100, 500, 20), torch.full((20,), 240.0)], 1).to(DEV)
    r = lib.pnp_ransac(line, lpix, K, num_iterations=256)
    assert torch.isfinite(r["transform"]).all() and 0 <= int(r["n_inlier"]) <= 20
