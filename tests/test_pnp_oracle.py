"""oracle/pnp_oracle.py (PnP-RANSAC, parity unpinned against OpenCV -- see its header): the P3P step solves its equations exactly, and the
whole estimate recovers known poses from noisy correspondences with 40 % outliers.  CPU only."""
import numpy as np

from oracle import pnp_oracle as po
from tests.helpers import pnp_scene


def test_p3p_solves_the_three_point_problem():
    X, px, K, T, _ = pnp_scene(3, n=60, outliers=0.0, noise=0.0)
    uv = px[:, ::-1].astype(np.float64)
    rng = np.random.default_rng(0)
    hit = 0
    for _ in range(100):
        idx = rng.choice(60, 3, replace=False)
        b = np.concatenate([uv[idx], np.ones((3, 1))], 1) @ np.linalg.inv(K).T
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        sols = po.p3p(X[idx].astype(np.float64), b)
        assert 1 <= len(sols) <= 4
        for R, t in sols:                                    # every solution reproduces the three pixels
            p, z = po.project(R, t, K, X[idx].astype(np.float64))
            assert (z > 0).all() and np.abs(p - uv[idx]).max() < 5e-2          # (float32 points / pixels; near-degenerate triples amplify)
        hit += any(np.abs(R - T[:3, :3]).max() < 1e-3 and np.abs(t - T[:3, 3]).max() < 3e-3 for R, t in sols)
    assert hit >= 97


def test_pnp_ransac_recovers_the_pose():
    for seed in (1, 2, 3):
        X, px, K, T, good = pnp_scene(seed)
        r = po.pnp_ransac(X, px, K, num_iterations=400, distance_tolerance=8.0, seed=seed)
        assert r["n_inlier"] >= 0.95 * good.sum()
        assert np.abs(r["transform"][:3, :3] - T[:3, :3]).max() < 3e-3 and np.abs(r["transform"][:3, 3] - T[:3, 3]).max() < 1e-2
    assert po.pnp_ransac(X[:3], px[:3], K) is None
