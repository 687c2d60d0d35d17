"""ORACLE -- CPU restatement of the PnP-RANSAC registration behind the 2D-3D fine matching.  TEST INFRASTRUCTURE.

PARITY UNPINNED: the reference calls OpenCV (`cv2.solvePnPRansac(..., iterationsCount=50000, reprojectionError=8.0, flags=cv2.SOLVEPNP_P3P)`,
Diff-Reg-2d3d/vision3d/utils/opencv.py:10-63, used by EXP/eval.py:174-182); cv2 is not installed in this image and its sources are not part of
/root/reference, so neither this oracle nor the kernel can be checked against it.  What is restated is the published algorithm of that call
(OpenCV calib3d/solvepnp.cpp): RANSAC over 4-point samples -- a P3P solution of the first three correspondences disambiguated by the fourth --
scored by the number of correspondences whose reprojection error is below the tolerance, then a refit on the inliers of the best model.
Stated differences: (i) all `num_iterations` hypotheses are scored (OpenCV stops early at 99 % confidence), drawn from the counter-based stream
diffreg_hip.synth.hash_bits instead of cv::RNG; (ii) the P3P step is Grunert's quartic (derived symbolically, see p3p()) instead of OpenCV's
Gao / Ke-Roumeliotis solvers -- the same solution set; (iii) the refit is a Gauss-Newton minimisation of the reprojection error over the
inliers started from the best hypothesis (OpenCV: EPnP on the inliers).  Only tests/ may import this module.
"""
import numpy as np

from diffreg_hip import synth


def p3p(P, J):
    """P [3,3] world points, J [3,3] unit bearing vectors of their pixels -> list of (R, t) with  lambda_i J_i = R P_i + t, lambda_i > 0.
    Distances s_i = |R P_i + t| from  s_j^2 + s_k^2 - 2 s_j s_k cos(J_j, J_k) = |P_j - P_k|^2;  with u = s2 / s1, v = s3 / s1 the resultant of the two
    quadratics in u is a quartic in v (coefficients generated with sympy: resultant(F1, F2, u), F1 = b2 (1 + u^2 - 2 u cg) - c2 (1 + v^2 - 2 v cb),
    F2 = b2 (u^2 + v^2 - 2 u v ca) - a2 (1 + v^2 - 2 v cb), common factor b2^2 dropped), u follows linearly from F1 - F2."""
    a2, b2, c2 = ((P[1] - P[2]) ** 2).sum(), ((P[0] - P[2]) ** 2).sum(), ((P[0] - P[1]) ** 2).sum()
    ca, cb, cg = J[1] @ J[2], J[0] @ J[2], J[0] @ J[1]
    A4 = a2 ** 2 - 2 * a2 * b2 - 2 * a2 * c2 + b2 ** 2 - 4 * b2 * c2 * ca ** 2 + 2 * b2 * c2 + c2 ** 2
    A3 = -4 * (a2 ** 2 * cb - a2 * b2 * ca * cg - a2 * b2 * cb - 2 * a2 * c2 * cb + b2 ** 2 * ca * cg - 2 * b2 * c2 * ca ** 2 * cb - b2 * c2 * ca * cg + b2 * c2 * cb
               + c2 ** 2 * cb)
    A2 = 2 * (2 * a2 ** 2 * cb ** 2 + a2 ** 2 - 4 * a2 * b2 * ca * cb * cg - 2 * a2 * b2 * cg ** 2 - 4 * a2 * c2 * cb ** 2 - 2 * a2 * c2 + 2 * b2 ** 2 * ca ** 2
              + 2 * b2 ** 2 * cg ** 2 - b2 ** 2 - 2 * b2 * c2 * ca ** 2 - 4 * b2 * c2 * ca * cb * cg + 2 * c2 ** 2 * cb ** 2 + c2 ** 2)
    A1 = -4 * (a2 ** 2 * cb - a2 * b2 * ca * cg - 2 * a2 * b2 * cb * cg ** 2 + a2 * b2 * cb - 2 * a2 * c2 * cb + b2 ** 2 * ca * cg - b2 * c2 * ca * cg - b2 * c2 * cb
               + c2 ** 2 * cb)
    A0 = a2 ** 2 - 4 * a2 * b2 * cg ** 2 + 2 * a2 * b2 - 2 * a2 * c2 + b2 ** 2 - 2 * b2 * c2 + c2 ** 2
    sols = []
    for v in np.roots([A4, A3, A2, A1, A0]):
        if abs(v.imag) > 1e-8 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        den = 2 * b2 * (ca * v - cg)
        if abs(den) < 1e-14:
            continue
        u = (2 * a2 * cb * v - a2 * v * v - a2 + b2 * v * v - b2 - 2 * c2 * cb * v + c2 * v * v + c2) / den
        if u <= 0:
            continue
        d = 1 + v * v - 2 * v * cb
        if d <= 0:
            continue
        s1 = np.sqrt(b2 / d)
        Q = np.stack([s1 * J[0], u * s1 * J[1], v * s1 * J[2]])
        sols.append(rigid_from_triangles(P, Q))
    return sols


def rigid_from_triangles(P, Q):
    """R, t with Q_i = R P_i + t for two congruent triangles: orthonormal frames of (P2 - P1, P3 - P1) and (Q2 - Q1, Q3 - Q1)"""
    def frame(X):
        e1 = X[1] - X[0]; e1 = e1 / np.linalg.norm(e1)
        e3 = np.cross(e1, X[2] - X[0]); e3 = e3 / np.linalg.norm(e3)
        return np.stack([e1, np.cross(e3, e1), e3], 1)
    R = frame(Q) @ frame(P).T
    return R, Q[0] - R @ P[0]


def project(R, t, K, X):
    Y = X @ R.T + t
    return np.stack([K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]], 1), Y[:, 2]


def refine(R, t, K, X, px, iters=10):
    """Gauss-Newton on the reprojection error, left perturbation exp([w]x) R"""
    for _ in range(iters):
        Y = X @ R.T + t
        x, y, z = Y[:, 0], Y[:, 1], Y[:, 2]
        fx, fy = K[0, 0], K[1, 1]
        r = np.stack([fx * x / z + K[0, 2] - px[:, 0], fy * y / z + K[1, 2] - px[:, 1]], 1).reshape(-1)
        Jp = np.zeros((len(X), 2, 3))
        Jp[:, 0, 0] = fx / z; Jp[:, 0, 2] = -fx * x / z ** 2; Jp[:, 1, 1] = fy / z; Jp[:, 1, 2] = -fy * y / z ** 2
        # d(Y)/d(w) = -[Y]x, d(Y)/d(t) = I
        Yx = np.zeros((len(X), 3, 3))
        Yx[:, 0, 1] = -z; Yx[:, 0, 2] = y; Yx[:, 1, 0] = z; Yx[:, 1, 2] = -x; Yx[:, 2, 0] = -y; Yx[:, 2, 1] = x
        Jm = np.concatenate([-(Jp @ Yx), Jp], 2).reshape(-1, 6)
        H, g = Jm.T @ Jm, Jm.T @ r
        d = np.linalg.solve(H + 1e-9 * np.eye(6), -g)
        w, th = d[:3], np.linalg.norm(d[:3])
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        dR = np.eye(3) + Wx + 0.5 * Wx @ Wx if th < 1e-8 else np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx
        R, t = dR @ R, dR @ t + d[3:]
    return R, t


def pnp_ransac(points, pixels, K, num_iterations=50000, distance_tolerance=8.0, seed=0, transposed=True):
    """-> dict(transform [4,4] (3D -> camera), n_inlier, best_iter) or None with fewer than 4 correspondences (opencv.py:36-38).
    pixels are (h, w) rows when `transposed` (opencv.py:42-43)."""
    X = np.asarray(points, dtype=np.float64)
    px = np.asarray(pixels, dtype=np.float64)
    if transposed:
        px = px[:, ::-1]
    n = len(X)
    if n < 4:
        return None
    K = np.asarray(K, dtype=np.float64)
    Kinv = np.linalg.inv(K)
    bear = np.concatenate([px, np.ones((n, 1))], 1) @ Kinv.T
    bear /= np.linalg.norm(bear, axis=1, keepdims=True)
    draw = (synth.hash_bits(seed, 0, num_iterations * 4) % np.uint64(n)).astype(np.int64).reshape(num_iterations, 4)
    best = (-1, None, None, -1)
    for it in range(num_iterations):
        idx = draw[it]
        if len(set(idx.tolist())) < 4:
            continue
        cands = p3p(X[idx[:3]], bear[idx[:3]])
        pick, perr = None, np.inf
        for (R, t) in cands:
            uv, z = project(R, t, K, X[idx[3:4]])
            e = np.inf if z[0] <= 0 else np.linalg.norm(uv[0] - px[idx[3]])
            if e < perr:
                pick, perr = (R, t), e
        if pick is None:
            continue
        uv, z = project(pick[0], pick[1], K, X)
        inl = (z > 0) & (((uv - px) ** 2).sum(1) < distance_tolerance ** 2)
        if inl.sum() > best[0]:
            best = (int(inl.sum()), pick, inl, it)
    if best[1] is None:
        return None
    R, t = refine(best[1][0], best[1][1], K, X[best[2]], px[best[2]]) if best[0] >= 4 else best[1]
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return dict(transform=T, n_inlier=best[0], best_iter=best[3], inliers=best[2])
