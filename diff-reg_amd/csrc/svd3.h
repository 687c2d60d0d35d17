// svd3.h -- fp64 3x3 one-sided Jacobi SVD shared by the Procrustes fit (procrustes.hip) and the
// correspondence RANSAC (metrics.hip).
#pragma once
#include "common.h"

namespace dr {

// One-sided Jacobi SVD of a 3x3 matrix, A = U diag(S) V^T, S descending.  Every index is static so the
// matrices stay in registers (a dynamically indexed local array would live in scratch).
// v_rcp_f64 / v_rsq_f64 seeds (~2^-26 accurate... at least 20 bits) + two Newton steps: ~1e-16 relative, a
// fraction of the instruction count of the IEEE-exact division / sqrt expansions (this code runs on ONE lane)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}

// floor2: column pairs with |a_p|^2 |a_q|^2 <= floor2 are left alone (a numerically zero column of a rank-2 matrix
// scaled to O(1) would otherwise keep `off` at noise level and burn all 30 sweeps); 0 = never skip
__device__ __forceinline__ void jacobi_rotate(double (&A)[3][3], double (&V)[3][3], const int p, const int q, double& off,
                                              const double floor2 = 0.0) {
    double al = 0, be = 0, ga = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        al += A[i][p] * A[i][p];
        be += A[i][q] * A[i][q];
        ga += A[i][p] * A[i][q];
    }
    if (fabs(ga) <= 1e-300 || al * be <= floor2) return;
    off = fmax(off, fabs(ga) * fast_rsqrt(al * be + 1e-300));
    const double zeta = (be - al) * 0.5 * fast_rcp(ga);
    const double z2 = 1.0 + zeta * zeta;
    const double tt = (zeta >= 0 ? 1.0 : -1.0) * fast_rcp(fabs(zeta) + z2 * fast_rsqrt(z2));
    const double c = fast_rsqrt(1.0 + tt * tt), s = c * tt;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ap = A[i][p], aq = A[i][q];
        A[i][p] = c * ap - s * aq;
        A[i][q] = s * ap + c * aq;
        const double vp = V[i][p], vq = V[i][q];
        V[i][p] = c * vp - s * vq;
        V[i][q] = s * vp + c * vq;
    }
}

__device__ __forceinline__ void swap_cols(double (&A)[3][3], double (&V)[3][3], double (&S)[3], const int a, const int b) {
    if (S[b] > S[a]) {
        const double ts = S[a]; S[a] = S[b]; S[b] = ts;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double ta = A[i][a]; A[i][a] = A[i][b]; A[i][b] = ta;
            const double tv = V[i][a]; V[i][a] = V[i][b]; V[i][b] = tv;
        }
    }
}

__device__ __forceinline__ void svd3_jacobi(double (&A)[3][3], double (&U)[3][3], double (&S)[3], double (&V)[3][3],
                                            const double floor2 = 0.0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        jacobi_rotate(A, V, 0, 1, off, floor2);
        jacobi_rotate(A, V, 0, 2, off, floor2);
        jacobi_rotate(A, V, 1, 2, off, floor2);
        if (off < 1e-12) break;        // quadratic convergence: the next sweep would be ~1e-24
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) S[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    swap_cols(A, V, S, 0, 1);
    swap_cols(A, V, S, 0, 2);
    swap_cols(A, V, S, 1, 2);
    const double tiny = 1e-200;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i][j] = S[j] > tiny ? A[i][j] / S[j] : (i == j ? 1.0 : 0.0);
    if (S[2] > 1e-14 * S[0] && S[2] > tiny) {
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i][2] = A[i][2] / S[2];
    } else {   // rank deficient: complete the basis (cond = inf/huge rejects it unless the gate is open)
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}

__device__ __forceinline__ double det3(const double (&m)[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

}  // namespace dr
