// procrustes_bwd.hip -- d loss / d conf of SoftProcrustesLayer.forward (3D/models/procrustes.py:17-93) on the device (second half).
// The K selected confidences w are the weights of the fit (the selection itself is piecewise constant):
//   wn = w / (sum |w| + 1e-4),  mx = sum wn X,  my = sum wn Y,  S = (Y - my)^T diag(wn) (X - mx),  S = U D V^T,
//   R = U diag(1, 1, det U det V) V^T,  t = my - R mx
// With gR, gt given:  gR' = gR - gt mx^T;  gU = gR' V F,  gV = gR'^T U F  (F = the constant sign matrix);
//   gS = U [ (E o (U^T gU - gU^T U)) D + D (E o (V^T gV - gV^T V)) ] V^T,  E_ij = 1 / (d_j^2 - d_i^2)   (the SVD's adjoint for square matrices)
//   g_my = gt - (1 - c) gS mx,  g_mx = -R^T gt - (1 - c) gS^T my,  c = sum wn
//   g_wn_k = (Y_k - my)^T gS (X_k - mx) + g_mx . X_k + g_my . Y_k;   g_w_k = g_wn_k / Z - sign(w_k) sum_j g_wn_j w_j / Z^2,  Z = sum |w| + 1e-4
// (checked against torch autograd through the reference's arithmetic to 1e-14 before it was written as a kernel; float64 throughout, like the
// reference's `.cpu().double().svd()`.)  One workgroup per pair; fixed-order reductions: bit-reproducible.
#include "kernels.h"
#include "svd3.h"

namespace dr {

struct ProcBwdArgs {
    const float* conf; const float* src_pcd; const float* tgt_pcd; const int* idx;   // idx [P, K] flat indices i M + j of the selected entries
    const int* kcount;                                                             // [P] entries of a pair that carry weight (4D: K from the mask sums) or null
    const float* gR; const float* gt;                                              // [P, 9], [P, 3]
    float* gconf;                                                                  // [P, N M], zeroed by the launcher
    int N, M, K;
};

__device__ __forceinline__ double blk_sum(double v, double* s_red, int t) {
    v = wave_sum(v);
    __syncthreads();
    if ((t & 63) == 0) s_red[t >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int k = 0; k < 4; ++k) r += s_red[k];
    return r;
}

__global__ __launch_bounds__(256) void procrustes_backward_kernel(ProcBwdArgs A) {
    __shared__ double s_red[4];
    __shared__ double s_m[16 + 9 + 6];            // gS [9] | g_mx [3] | g_my [3] | mx [3] | my [3]
    const int pair = blockIdx.x, t = threadIdx.x, N = A.N, M = A.M, K = A.K;
    const float* conf = A.conf + (size_t)pair * N * M;
    const float* ps = A.src_pcd + (size_t)pair * N * 3;
    const float* pt = A.tgt_pcd + (size_t)pair * M * 3;
    const int* idx = A.idx + (size_t)pair * K;
    const int kmax = A.kcount ? min(K, A.kcount[pair]) : K;
    auto weight = [&](int k) -> double { return k < kmax ? (double)conf[idx[k]] : 0.0; };
    // ---- moments: Z, sum w X, sum w Y, sum w Y X^T, sum w (17 block reductions of K terms; K <= 4096)
    double acc[17];                                // |w|, w X [3], w Y [3], w Y X^T [9], w
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.0;
    for (int k = t; k < kmax; k += 256) {          // (entries beyond a pair's own count carry no weight and their indices are unspecified)
        const double w = weight(k);
        const int e = idx[k], i = e / M, j = e % M;
        const double X[3] = {ps[i * 3], ps[i * 3 + 1], ps[i * 3 + 2]}, Y[3] = {pt[j * 3], pt[j * 3 + 1], pt[j * 3 + 2]};
        acc[0] += fabs(w);
        acc[16] += w;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[1 + a] += w * X[a];
            acc[4 + a] += w * Y[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[7 + 3 * a + b] += w * Y[a] * X[b];
        }
    }
    double sum[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) sum[i] = blk_sum(acc[i], s_red, t);
    const double Z = sum[0] + 1e-4, c = sum[16] / Z;
    if (t == 0) {
        double mx[3], my[3], S[3][3], U[3][3], V[3][3], D[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mx[a] = sum[1 + a] / Z; my[a] = sum[4 + a] / Z; }
        // S = sum wn (Y - my)(X - mx)^T = sum wn Y X^T - my (sum wn X)^T - (sum wn Y) mx^T + c my mx^T = sum wn Y X^T - (2 - c) my mx^T
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) S[a][b] = sum[7 + 3 * a + b] / Z - (2.0 - c) * my[a] * mx[b];
        double Aw[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Aw[a][b] = S[a][b];
        svd3_jacobi(Aw, U, D, V);
        const double dd = det3(U) * det3(V);
        const double F[3] = {1.0, 1.0, dd};
        double R[3][3], gRp[3][3], gU[3][3], gV[3][3];
        const float* gR = A.gR + (size_t)pair * 9; const float* gt = A.gt + (size_t)pair * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                R[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + dd * U[a][2] * V[b][2];
                gRp[a][b] = (double)gR[a * 3 + b] - (double)gt[a] * mx[b];
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double u = 0, v = 0;
#pragma unroll
                for (int b = 0; b < 3; ++b) { u += gRp[a][b] * V[b][j]; v += gRp[b][a] * U[b][j]; }
                gU[a][j] = u * F[j]; gV[a][j] = v * F[j];
            }
        // A1 = E o (U^T gU - gU^T U), A2 = E o (V^T gV - gV^T V);  inner = A1 D + D A2
        double inner[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double a1 = 0, a2 = 0;
                if (i != j) {
                    double ug = 0, gu = 0, vg = 0, gv = 0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) { ug += U[q][i] * gU[q][j]; gu += gU[q][i] * U[q][j]; vg += V[q][i] * gV[q][j]; gv += gV[q][i] * V[q][j]; }
                    const double den = D[j] * D[j] - D[i] * D[i];
                    const double E = fabs(den) > 1e-300 ? 1.0 / den : 0.0;        // (equal singular values: the fit's rotation is not unique there)
                    a1 = E * (ug - gu); a2 = E * (vg - gv);
                }
                inner[i][j] = a1 * D[j] + D[i] * a2;
            }
        double gS[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                double v = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) v += U[a][i] * inner[i][j] * V[b][j];
                gS[a][b] = v;
                s_m[3 * a + b] = v;
            }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double gmx = 0, gmy = (double)gt[b];
#pragma unroll
            for (int a = 0; a < 3; ++a) { gmx += -R[a][b] * (double)gt[a] - (1.0 - c) * gS[a][b] * my[a]; gmy += -(1.0 - c) * gS[b][a] * mx[a]; }
            s_m[9 + b] = gmx; s_m[12 + b] = gmy; s_m[15 + b] = mx[b]; s_m[18 + b] = my[b];
        }
    }
    __syncthreads();
    double gS[3][3], gmx[3], gmy[3], mx[3], my[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gmx[a] = s_m[9 + a]; gmy[a] = s_m[12 + a]; mx[a] = s_m[15 + a]; my[a] = s_m[18 + a];
#pragma unroll
        for (int b = 0; b < 3; ++b) gS[a][b] = s_m[3 * a + b];
    }
    auto gwn = [&](int k) -> double {
        const int e = idx[k], i = e / M, j = e % M;
        const double X[3] = {ps[i * 3], ps[i * 3 + 1], ps[i * 3 + 2]}, Y[3] = {pt[j * 3], pt[j * 3 + 1], pt[j * 3 + 2]};
        double v = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            v += gmx[a] * X[a] + gmy[a] * Y[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) v += (Y[a] - my[a]) * gS[a][b] * (X[b] - mx[b]);
        }
        return v;
    };
    double dot = 0;
    for (int k = t; k < kmax; k += 256) dot += gwn(k) * weight(k);
    dot = blk_sum(dot, s_red, t);
    float* g = A.gconf + (size_t)pair * N * M;
    for (int k = t; k < kmax; k += 256) {
        const double w = weight(k);
        const double sgn = w > 0 ? 1.0 : (w < 0 ? -1.0 : 0.0);
        g[idx[k]] = (float)(gwn(k) / Z - sgn * dot / (Z * Z));
    }
}

int launch_procrustes_backward(const ProcBwdArgs& a, int P, hipStream_t st) {
    if (P <= 0) return DR_OK;
    DR_HIP_CHECK(hipMemsetAsync(a.gconf, 0, (size_t)P * a.N * a.M * sizeof(float), st));
    if (a.K <= 0) return DR_OK;
    hipLaunchKernelGGL(procrustes_backward_kernel, dim3(P), dim3(256), 0, st, a);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" int dr_procrustes_backward_f32(int P, int N, int M, int K, const float* conf, const float* src_pcd, const float* tgt_pcd, const int32_t* topk_idx,
                                          const int32_t* k_count, const float* grad_R, const float* grad_t, float* grad_conf, void* stream) {
    if (P < 0 || N < 1 || M < 1 || K < 0 || !conf || !src_pcd || !tgt_pcd || (K > 0 && !topk_idx) || !grad_R || !grad_t || !grad_conf) return DR_EINVAL;
    ProcBwdArgs a;
    a.conf = conf; a.src_pcd = src_pcd; a.tgt_pcd = tgt_pcd; a.idx = topk_idx; a.kcount = k_count; a.gR = grad_R; a.gt = grad_t; a.gconf = grad_conf;
    a.N = N; a.M = M; a.K = K;
    return launch_procrustes_backward(a, P, (hipStream_t)stream);
}
