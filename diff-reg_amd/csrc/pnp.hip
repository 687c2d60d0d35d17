// pnp.hip -- PnP-RANSAC registration behind the 2D-3D fine matching: what EXP/eval.py:174-182 gets from
// vision3d.utils.opencv.registration_with_pnp_ransac (opencv.py:10-63) = cv2.solvePnPRansac(points, pixels, K, 0, iterationsCount = 50000,
// reprojectionError = 8.0, flags = SOLVEPNP_P3P).  OpenCV is not part of the reference tree: this is the published algorithm of that call
// (calib3d/solvepnp.cpp: RANSAC over 4-point samples, P3P on three + disambiguation by the fourth, inlier count under the reprojection
// tolerance, refit on the inliers), restated in oracle/pnp_oracle.py with the differences listed there (all hypotheses scored, counter-based
// sampling, Grunert's quartic for P3P, Gauss-Newton refit).  PARITY UNPINNED against OpenCV (absent); pinned against the oracle.
//
// Kernel 1: one thread per hypothesis (float64: P3P, Durand-Kerner roots of the quartic), the correspondences streamed through LDS so that a
// workgroup's 256 hypotheses score the same point at the same time; per-workgroup best (most inliers, lowest hypothesis index).
// Kernel 2: one workgroup: best over the workgroups, the inlier set, 10 Gauss-Newton steps on the reprojection error (normal equations
// accumulated in a fixed order), the 4 x 4 transform.
#include "kernels.h"

namespace dr {
namespace {

__host__ __device__ __forceinline__ uint64_t pnp_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t pnp_hash(uint64_t base, uint64_t idx) { return pnp_mix64(pnp_mix64(idx ^ base) + base) >> 11; }

struct PnpModel { double R[9]; double t[3]; };
struct PnpBest { int count; int it; PnpModel m; };

struct PnpArgs {
    const float* X; const float* px; int n; int transposed; double fx, fy, cx, cy; int iters; uint64_t base; double tol2;
    PnpBest* blk; int nblk;
    double* T; int* n_inlier; int* best_iter; uint8_t* inlier_mask;
};

__device__ __forceinline__ void load_px(const PnpArgs& A, int i, double& u, double& v) {
    const double a = A.px[2 * i], b = A.px[2 * i + 1];
    u = A.transposed ? b : a;       // (h, w) rows -> (w, h)  (opencv.py:42-43)
    v = A.transposed ? a : b;
}

// R, t with Q_i = R P_i + t for two congruent triangles (orthonormal frames of their first two edges)
__device__ __forceinline__ bool rigid3(const double (&P)[3][3], const double (&Q)[3][3], PnpModel& m) {
    double Fp[3][3], Fq[3][3];
    auto frame = [](const double (&X)[3][3], double (&F)[3][3]) -> bool {
        double e1[3] = {X[1][0] - X[0][0], X[1][1] - X[0][1], X[1][2] - X[0][2]};
        double d[3] = {X[2][0] - X[0][0], X[2][1] - X[0][1], X[2][2] - X[0][2]};
        double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
        if (!(n1 > 1e-12)) return false;
        e1[0] /= n1; e1[1] /= n1; e1[2] /= n1;
        double e3[3] = {e1[1] * d[2] - e1[2] * d[1], e1[2] * d[0] - e1[0] * d[2], e1[0] * d[1] - e1[1] * d[0]};
        double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
        if (!(n3 > 1e-12)) return false;
        e3[0] /= n3; e3[1] /= n3; e3[2] /= n3;
        double e2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
        for (int r = 0; r < 3; ++r) { F[r][0] = e1[r]; F[r][1] = e2[r]; F[r][2] = e3[r]; }
        return true;
    };
    if (!frame(P, Fp) || !frame(Q, Fq)) return false;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m.R[3 * r + c] = Fq[r][0] * Fp[c][0] + Fq[r][1] * Fp[c][1] + Fq[r][2] * Fp[c][2];
    for (int r = 0; r < 3; ++r) m.t[r] = Q[0][r] - (m.R[3 * r] * P[0][0] + m.R[3 * r + 1] * P[0][1] + m.R[3 * r + 2] * P[0][2]);
    return true;
}

// real positive roots of A4 v^4 + .. + A0 by Durand-Kerner (complex float64, fixed iteration count) + two Newton steps on the real part
__device__ __forceinline__ int quartic_real_roots(double A4, double A3, double A2, double A1, double A0, double (&out)[4]) {
    if (!(fabs(A4) > 1e-300)) return 0;
    const double a = A3 / A4, b = A2 / A4, c = A1 / A4, d = A0 / A4;
    // radius bound of the roots (Cauchy) for the start values
    const double rad = 1.0 + fmax(fmax(fabs(a), fabs(b)), fmax(fabs(c), fabs(d)));
    double zr[4], zi[4];
    for (int k = 0; k < 4; ++k) { const double ang = 0.4 + 1.5707963267948966 * k; zr[k] = 0.5 * rad * cos(ang); zi[k] = 0.5 * rad * sin(ang); }
    for (int it = 0; it < 60; ++it) {
        for (int k = 0; k < 4; ++k) {
            // p(z) by Horner
            double pr = 1.0, pi = 0.0;
            const double cf[4] = {a, b, c, d};
            for (int j = 0; j < 4; ++j) { const double nr = pr * zr[k] - pi * zi[k] + cf[j], ni = pr * zi[k] + pi * zr[k]; pr = nr; pi = ni; }
            double qr = 1.0, qi = 0.0;
            for (int j = 0; j < 4; ++j)
                if (j != k) { const double dr_ = zr[k] - zr[j], di = zi[k] - zi[j]; const double nr = qr * dr_ - qi * di, ni = qr * di + qi * dr_; qr = nr; qi = ni; }
            const double den = qr * qr + qi * qi;
            if (den > 0) { zr[k] -= (pr * qr + pi * qi) / den; zi[k] -= (pi * qr - pr * qi) / den; }
        }
    }
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        if (fabs(zi[k]) > 1e-8 * fmax(1.0, fabs(zr[k])) || !(zr[k] > 0)) continue;
        double v = zr[k];
        for (int s = 0; s < 2; ++s) {
            const double f = (((v + a) * v + b) * v + c) * v + d, fp = ((4 * v + 3 * a) * v + 2 * b) * v + c;
            if (fabs(fp) > 1e-300) v -= f / fp;
        }
        if (v > 0) out[n++] = v;
    }
    return n;
}

__device__ __forceinline__ bool reproject(const PnpModel& m, const PnpArgs& A, double X, double Y, double Z, double& u, double& v) {
    const double x = m.R[0] * X + m.R[1] * Y + m.R[2] * Z + m.t[0], y = m.R[3] * X + m.R[4] * Y + m.R[5] * Z + m.t[1],
                 z = m.R[6] * X + m.R[7] * Y + m.R[8] * Z + m.t[2];
    if (!(z > 0)) return false;
    u = A.fx * x / z + A.cx; v = A.fy * y / z + A.cy;
    return true;
}

constexpr int PNP_CHUNK = 512;

__global__ __launch_bounds__(256) void pnp_hyp_kernel(PnpArgs A) {
    __shared__ float sX[PNP_CHUNK][3];
    __shared__ float sP[PNP_CHUNK][2];
    __shared__ int s_cnt[256];
    const int it = blockIdx.x * 256 + threadIdx.x;
    PnpModel m;
    bool have = false;
    if (it < A.iters) {
        int idx[4];
        for (int s = 0; s < 4; ++s) idx[s] = (int)(pnp_hash(A.base, (uint64_t)it * 4 + s) % (uint64_t)A.n);
        const bool distinct = idx[0] != idx[1] && idx[0] != idx[2] && idx[0] != idx[3] && idx[1] != idx[2] && idx[1] != idx[3] && idx[2] != idx[3];
        if (distinct) {
            double P[3][3], J[3][3];
            for (int s = 0; s < 3; ++s) {
                for (int c = 0; c < 3; ++c) P[s][c] = A.X[3 * idx[s] + c];
                double u, v;
                load_px(A, idx[s], u, v);
                const double bx = (u - A.cx) / A.fx, by = (v - A.cy) / A.fy, nb = sqrt(bx * bx + by * by + 1.0);
                J[s][0] = bx / nb; J[s][1] = by / nb; J[s][2] = 1.0 / nb;
            }
            auto d2 = [&](int i, int j) { double s = 0; for (int c = 0; c < 3; ++c) { const double d = P[i][c] - P[j][c]; s += d * d; } return s; };
            auto dt = [&](int i, int j) { return J[i][0] * J[j][0] + J[i][1] * J[j][1] + J[i][2] * J[j][2]; };
            const double a2 = d2(1, 2), b2 = d2(0, 2), c2 = d2(0, 1), ca = dt(1, 2), cb = dt(0, 2), cg = dt(0, 1);
            const double A4 = a2 * a2 - 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 4 * b2 * c2 * ca * ca + 2 * b2 * c2 + c2 * c2;
            const double A3 = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - 2 * b2 * c2 * ca * ca * cb - b2 * c2 * ca * cg +
                                   b2 * c2 * cb + c2 * c2 * cb);
            const double A2 = 2 * (2 * a2 * a2 * cb * cb + a2 * a2 - 4 * a2 * b2 * ca * cb * cg - 2 * a2 * b2 * cg * cg - 4 * a2 * c2 * cb * cb - 2 * a2 * c2 +
                                   2 * b2 * b2 * ca * ca + 2 * b2 * b2 * cg * cg - b2 * b2 - 2 * b2 * c2 * ca * ca - 4 * b2 * c2 * ca * cb * cg + 2 * c2 * c2 * cb * cb + c2 * c2);
            const double A1 = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - 2 * a2 * b2 * cb * cg * cg + a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - b2 * c2 * ca * cg -
                                   b2 * c2 * cb + c2 * c2 * cb);
            const double A0 = a2 * a2 - 4 * a2 * b2 * cg * cg + 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 2 * b2 * c2 + c2 * c2;
            double roots[4];
            const int nr = quartic_real_roots(A4, A3, A2, A1, A0, roots);
            double X4[3] = {A.X[3 * idx[3]], A.X[3 * idx[3] + 1], A.X[3 * idx[3] + 2]}, u4, v4;
            load_px(A, idx[3], u4, v4);
            double best_e = INFINITY;
            for (int r = 0; r < nr; ++r) {
                const double v = roots[r], den = 2 * b2 * (ca * v - cg);
                if (!(fabs(den) > 1e-14)) continue;
                const double u = (2 * a2 * cb * v - a2 * v * v - a2 + b2 * v * v - b2 - 2 * c2 * cb * v + c2 * v * v + c2) / den;
                const double dd = 1 + v * v - 2 * v * cb;
                if (!(u > 0) || !(dd > 0)) continue;
                const double s1 = sqrt(b2 / dd);
                double Q[3][3];
                for (int c = 0; c < 3; ++c) { Q[0][c] = s1 * J[0][c]; Q[1][c] = u * s1 * J[1][c]; Q[2][c] = v * s1 * J[2][c]; }
                PnpModel cand;
                if (!rigid3(P, Q, cand)) continue;
                double pu, pv;
                if (!reproject(cand, A, X4[0], X4[1], X4[2], pu, pv)) continue;
                const double e = (pu - u4) * (pu - u4) + (pv - v4) * (pv - v4);
                if (e < best_e) { best_e = e; m = cand; have = true; }
            }
        }
    }
    int cnt = 0;
    for (int c0 = 0; c0 < A.n; c0 += PNP_CHUNK) {
        __syncthreads();
        for (int e = threadIdx.x; e < PNP_CHUNK; e += 256) {
            const int i = c0 + e;
            if (i < A.n) {
                sX[e][0] = A.X[3 * i]; sX[e][1] = A.X[3 * i + 1]; sX[e][2] = A.X[3 * i + 2];
                double u, v;
                load_px(A, i, u, v);
                sP[e][0] = (float)u; sP[e][1] = (float)v;
            }
        }
        __syncthreads();
        if (have) {
            const int lim = min(PNP_CHUNK, A.n - c0);
            for (int e = 0; e < lim; ++e) {
                double pu, pv;
                if (reproject(m, A, sX[e][0], sX[e][1], sX[e][2], pu, pv)) {
                    const double du = pu - (double)sP[e][0], dv = pv - (double)sP[e][1];
                    cnt += (du * du + dv * dv < A.tol2) ? 1 : 0;
                }
            }
        }
    }
    s_cnt[threadIdx.x] = have ? cnt : -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        int bc = -1, bt = 0;
        for (int k = 0; k < 256; ++k)
            if (s_cnt[k] > bc) { bc = s_cnt[k]; bt = k; }
        A.blk[blockIdx.x].count = bc;
        A.blk[blockIdx.x].it = blockIdx.x * 256 + bt;
        s_cnt[0] = bt;
    }
    __syncthreads();
    if ((int)threadIdx.x == s_cnt[0] && have) A.blk[blockIdx.x].m = m;
}

__global__ __launch_bounds__(256) void pnp_final_kernel(PnpArgs A) {
    __shared__ PnpModel sm;
    __shared__ double s_part[4][27];
    __shared__ int s_best[2];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) {
        int bc = -1, bi = -1;
        for (int k = 0; k < A.nblk; ++k)
            if (A.blk[k].count > bc) { bc = A.blk[k].count; bi = k; }
        s_best[0] = bc; s_best[1] = bi;
        if (bi >= 0) sm = A.blk[bi].m;
    }
    __syncthreads();
    const int bc = s_best[0], bi = s_best[1];
    if (bc < 0) {
        if (t == 0) { *A.n_inlier = 0; *A.best_iter = -1; for (int k = 0; k < 16; ++k) A.T[k] = (k % 5 == 0) ? 1.0 : 0.0; }
        return;
    }
    // the inlier set of the best hypothesis
    for (int i = t; i < A.n; i += 256) {
        double pu, pv, u, v;
        load_px(A, i, u, v);
        bool in = false;
        if (reproject(sm, A, A.X[3 * i], A.X[3 * i + 1], A.X[3 * i + 2], pu, pv)) in = ((pu - u) * (pu - u) + (pv - v) * (pv - v)) < A.tol2;
        A.inlier_mask[i] = in ? 1 : 0;
    }
    __syncthreads();
    if (bc >= 4) {
        for (int gn = 0; gn < 10; ++gn) {
            double acc[27];
            for (int k = 0; k < 27; ++k) acc[k] = 0.0;
            for (int i = t; i < A.n; i += 256) {
                if (!A.inlier_mask[i]) continue;
                const double X = A.X[3 * i], Y = A.X[3 * i + 1], Z = A.X[3 * i + 2];
                const double x = sm.R[0] * X + sm.R[1] * Y + sm.R[2] * Z + sm.t[0], y = sm.R[3] * X + sm.R[4] * Y + sm.R[5] * Z + sm.t[1],
                             z = sm.R[6] * X + sm.R[7] * Y + sm.R[8] * Z + sm.t[2];
                double u, v;
                load_px(A, i, u, v);
                const double r0 = A.fx * x / z + A.cx - u, r1 = A.fy * y / z + A.cy - v;
                // rows of the Jacobian w.r.t. (w, t): Jp = [[fx/z, 0, -fx x/z^2], [0, fy/z, -fy y/z^2]], dY/dw = -[Y]x
                const double a0 = A.fx / z, a2 = -A.fx * x / (z * z), b1 = A.fy / z, b2 = -A.fy * y / (z * z);
                // -Jp [Y]x with [Y]x = [[0,-z,y],[z,0,-x],[-y,x,0]]
                const double J0[6] = {-(a2 * (-y)), -(a0 * (-z) + a2 * x), -(a0 * y), a0, 0.0, a2};
                const double J1[6] = {-(b1 * z + b2 * (-y)), -(b2 * x), -(b1 * (-x)), 0.0, b1, b2};
                int k = 0;
                for (int p = 0; p < 6; ++p)
                    for (int q = p; q < 6; ++q) acc[k++] += J0[p] * J0[q] + J1[p] * J1[q];
                for (int p = 0; p < 6; ++p) acc[21 + p] += J0[p] * r0 + J1[p] * r1;
            }
            for (int k = 0; k < 27; ++k) { const double v = wave_sum(acc[k]); if (lane == 0) s_part[w][k] = v; }
            __syncthreads();
            if (t == 0) {
                double H[6][6], g[6];
                int k = 0;
                for (int p = 0; p < 6; ++p)
                    for (int q = p; q < 6; ++q) { const double v = (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k]); H[p][q] = H[q][p] = v; ++k; }
                for (int p = 0; p < 6; ++p) { g[p] = -((s_part[0][21 + p] + s_part[1][21 + p]) + (s_part[2][21 + p] + s_part[3][21 + p])); H[p][p] += 1e-9; }
                // Gaussian elimination with partial pivoting
                for (int c = 0; c < 6; ++c) {
                    int pv = c;
                    for (int r = c + 1; r < 6; ++r) if (fabs(H[r][c]) > fabs(H[pv][c])) pv = r;
                    if (pv != c) { for (int q = 0; q < 6; ++q) { const double tmp = H[c][q]; H[c][q] = H[pv][q]; H[pv][q] = tmp; } const double tg = g[c]; g[c] = g[pv]; g[pv] = tg; }
                    const double piv = H[c][c];
                    if (!(fabs(piv) > 1e-300)) continue;
                    for (int r = c + 1; r < 6; ++r) { const double f = H[r][c] / piv; for (int q = c; q < 6; ++q) H[r][q] -= f * H[c][q]; g[r] -= f * g[c]; }
                }
                double d[6];
                for (int c = 5; c >= 0; --c) { double s = g[c]; for (int q = c + 1; q < 6; ++q) s -= H[c][q] * d[q]; d[c] = fabs(H[c][c]) > 1e-300 ? s / H[c][c] : 0.0; }
                const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                const double ka = th < 1e-8 ? 1.0 : sin(th) / th, kb = th < 1e-8 ? 0.5 : (1 - cos(th)) / (th * th);
                const double W[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
                double dR[9];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        double w2 = 0;
                        for (int q = 0; q < 3; ++q) w2 += W[3 * r + q] * W[3 * q + c];
                        dR[3 * r + c] = (r == c ? 1.0 : 0.0) + ka * W[3 * r + c] + kb * w2;
                    }
                PnpModel nm;
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) nm.R[3 * r + c] = dR[3 * r] * sm.R[c] + dR[3 * r + 1] * sm.R[3 + c] + dR[3 * r + 2] * sm.R[6 + c];
                    nm.t[r] = dR[3 * r] * sm.t[0] + dR[3 * r + 1] * sm.t[1] + dR[3 * r + 2] * sm.t[2] + d[3 + r];
                }
                sm = nm;
            }
            __syncthreads();
        }
    }
    if (t == 0) {
        *A.n_inlier = bc;
        *A.best_iter = A.blk[bi].it;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) A.T[4 * r + c] = sm.R[3 * r + c]; A.T[4 * r + 3] = sm.t[r]; }
        A.T[12] = A.T[13] = A.T[14] = 0.0; A.T[15] = 1.0;
    }
}

}  // namespace
}  // namespace dr

extern "C" {

size_t dr_pnp_ransac_workspace_bytes(int n, int iters) {
    if (n < 0 || iters < 1) return 0;
    return (size_t)((iters + 255) / 256) * sizeof(dr::PnpBest) + (((size_t)n + 255) & ~(size_t)255);
}

int dr_pnp_ransac_f64(int n, const float* points, const float* pixels, int transposed, const double* intrinsics_host, int iters, double distance_tolerance,
                      uint64_t seed, double* transform, int32_t* n_inlier, int32_t* best_iter, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 4 || iters < 1 || !points || !pixels || !intrinsics_host || !transform || !n_inlier || !best_iter || !workspace) return DR_EINVAL;
    if (workspace_bytes < dr_pnp_ransac_workspace_bytes(n, iters)) return DR_EWORKSPACE;
    dr::PnpArgs A;
    A.X = points; A.px = pixels; A.n = n; A.transposed = transposed ? 1 : 0;
    A.fx = intrinsics_host[0]; A.cx = intrinsics_host[2]; A.fy = intrinsics_host[4]; A.cy = intrinsics_host[5];
    A.iters = iters; A.base = dr::pnp_mix64(seed * 0x100000001B3ull + 0ull); A.tol2 = distance_tolerance * distance_tolerance;
    A.nblk = (iters + 255) / 256;
    A.blk = (dr::PnpBest*)workspace;
    A.inlier_mask = (uint8_t*)workspace + (size_t)A.nblk * sizeof(dr::PnpBest);
    A.T = transform; A.n_inlier = n_inlier; A.best_iter = best_iter;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dr::pnp_hyp_kernel, dim3(A.nblk), dim3(256), 0, st, A);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::pnp_final_kernel, dim3(1), dim3(256), 0, st, A);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // extern "C"
