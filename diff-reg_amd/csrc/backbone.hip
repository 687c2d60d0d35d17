// backbone.hip -- the point-convolution side of the KPFCN backbone (3D/models/blocks.py).
//   kpconv_gather   neighbour gather + kernel-point influences + reduction over the neighbours of KPConv.forward
//                   (blocks.py:288-375, 390-393): wf[q][k*Cin + c] = (sum_h w[q][k][h] x[nb[q][h]][c]) / num_q, the A operand
//                   of ONE GEMM with the [Cout][K*Cin] view of the kernel weights (the reference multiplies per kernel
//                   point and sums over K: blocks.py:382-387)
//   col_stats / norm_apply   BatchNormBlock with use_bn = InstanceNorm1d over the points of the stacked cloud, per
//                   channel, no affine, biased variance, eps 1e-5 (blocks.py:430-446), fused with LeakyReLU(0.1) and
//                   with the residual sum of ResnetBottleneckBlock.forward (blocks.py:650-660)
//   gather_max / gather_rows  max_pool / closest_pool over index lists with a zero "shadow" row (blocks.py:56-87)
// All HBM-bound gathers / reductions; index lists are the reference's int64 tensors.
#include "kernels.h"

namespace dr {

constexpr int KP_MAXK = 16, KP_MAXH = 64;

// one wave per query point; lane l owns channels l, l + 64, .. (CPL of them); ONE pass over the neighbours' features:
// a neighbour's K influences are read once (4 x ds_read_b128 from the [h][16] image) and applied to all the lane's
// channels, and the per-neighbour feature sums of the normalisation are accumulated in the same pass
template <int CPL>
__global__ __launch_bounds__(256) void kpconv_gather_kernel(int Nq, int Ns, int H, int Cin, int K, const float* __restrict__ q_pts,
                                                            const float* __restrict__ s_pts, const long long* __restrict__ nb,
                                                            const float* __restrict__ x, const float* __restrict__ kp, float extent, int mode,
                                                            float* __restrict__ wf, int ldw) {
    __shared__ __attribute__((aligned(16))) float s_w[4][KP_MAXH * KP_MAXK];     // influences [h][k] of the wave's query
    __shared__ int s_idx[4][KP_MAXH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= Nq) return;
    const float qx = q_pts[q * 3], qy = q_pts[q * 3 + 1], qz = q_pts[q * 3 + 2];
    if (lane < H) {
        const long long id = nb[(size_t)q * H + lane];
        const bool shadow = id >= Ns || id < 0;
        s_idx[w][lane] = shadow ? -1 : (int)id;
        // the shadow point sits at +1e6 on every axis (blocks.py:288): its influence is 0
        const float nx = (shadow ? 1e6f : s_pts[id * 3]) - qx, ny = (shadow ? 1e6f : s_pts[id * 3 + 1]) - qy,
                    nz = (shadow ? 1e6f : s_pts[id * 3 + 2]) - qz;
        float d2[KP_MAXK];
        int kmin = 0;                                               // the neighbour's nearest kernel point (first minimum, as torch.argmin)
#pragma unroll
        for (int k = 0; k < KP_MAXK; ++k) {
            d2[k] = INFINITY;
            if (k < K) {
                const float dx = nx - kp[k * 3], dy = ny - kp[k * 3 + 1], dz = nz - kp[k * 3 + 2];
                d2[k] = dx * dx + dy * dy + dz * dz;
                if (d2[k] < d2[kmin]) kmin = k;
            }
        }
#pragma unroll
        for (int k = 0; k < KP_MAXK; ++k) {
            float wv = k < K ? kp_influence(mode, d2[k], extent) : 0.f;         // 'linear' in every shipped configuration (blocks.py:309-312)
            if ((mode & KP_CLOSEST) && k != kmin) wv = 0.f;
            s_w[w][lane * KP_MAXK + k] = wv;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float acc[CPL][KP_MAXK];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int k = 0; k < KP_MAXK; ++k) acc[c][k] = 0.f;
    int num = 0;
    for (int h = 0; h < H; ++h) {
        const int id = s_idx[w][h];
        if (id < 0) continue;                                       // zero features of the shadow row (blocks.py:369); uniform per wave
        float wk[KP_MAXK];
#pragma unroll
        for (int k4 = 0; k4 < KP_MAXK / 4; ++k4) {
            const float4 t4 = *reinterpret_cast<const float4*>(&s_w[w][h * KP_MAXK + 4 * k4]);
            wk[4 * k4] = t4.x; wk[4 * k4 + 1] = t4.y; wk[4 * k4 + 2] = t4.z; wk[4 * k4 + 3] = t4.w;
        }
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            const float xv = ch < Cin ? x[(size_t)id * Cin + ch] : 0.f;
            part += xv;
#pragma unroll
            for (int k = 0; k < KP_MAXK; ++k) acc[c][k] = fmaf(wk[k], xv, acc[c][k]);
        }
        // neighbours with a positive feature sum (blocks.py:390-392), summed over all channels in a fixed order
        part = wave_sum(part);
        num += part > 0.f ? 1 : 0;
    }
    const float inv = 1.f / (float)(num > 1 ? num : 1);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + 64 * c;
        if (ch < Cin)
#pragma unroll
            for (int k = 0; k < KP_MAXK; ++k)
                if (k < K) wf[(size_t)q * ldw + k * Cin + ch] = acc[c][k] * inv;
    }
    // zero the padding columns K*Cin .. ldw-1 (the GEMM reads whole float4s)
    for (int c = K * Cin + lane; c < ldw; c += 64) wf[(size_t)q * ldw + c] = 0.f;
}

int launch_kpconv_gather(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const long long* nb,
                         const float* x, const float* kp, float extent, float* wf, int ldw, hipStream_t st, int mode) {
    if (Nq <= 0) return DR_OK;
    if ((mode & 3) == 3 || (mode & ~7)) return DR_EINVAL;
    if (K > KP_MAXK || H > KP_MAXH || H < 1 || Cin < 1 || Cin > 512 || ldw < K * Cin) return DR_ENOSUP;
    const dim3 grid((Nq + 3) / 4), blk(256);
    if (Cin <= 64) hipLaunchKernelGGL(kpconv_gather_kernel<1>, grid, blk, 0, st, Nq, Ns, H, Cin, K, q_pts, s_pts, nb, x, kp, extent, mode, wf, ldw);
    else if (Cin <= 128) hipLaunchKernelGGL(kpconv_gather_kernel<2>, grid, blk, 0, st, Nq, Ns, H, Cin, K, q_pts, s_pts, nb, x, kp, extent, mode, wf, ldw);
    else if (Cin <= 256) hipLaunchKernelGGL(kpconv_gather_kernel<4>, grid, blk, 0, st, Nq, Ns, H, Cin, K, q_pts, s_pts, nb, x, kp, extent, mode, wf, ldw);
    else hipLaunchKernelGGL(kpconv_gather_kernel<8>, grid, blk, 0, st, Nq, Ns, H, Cin, K, q_pts, s_pts, nb, x, kp, extent, mode, wf, ldw);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---- per-channel statistics over the rows: stage 1 partial sums (double), stage 2 mean / rstd -------------------------
__global__ __launch_bounds__(256) void col_stats_partial_kernel(int N, int C, const float* __restrict__ x, int ldx, int rows_per,
                                                                double* __restrict__ part) {
    // block = 32 channels x 8 row lanes; grid = (ceil(C/32), R)
    __shared__ double s_s[8][32], s_q[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rows_per, r1 = min(N, r0 + rows_per);
    double s = 0.0, qq = 0.0;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += 8) {
            const double v = (double)x[(size_t)r * ldx + c];
            s += v; qq += v * v;
        }
    s_s[rl][cl] = s; s_q[rl][cl] = qq;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < 8; ++k) { s += s_s[k][cl]; qq += s_q[k][cl]; }
        part[((size_t)blockIdx.y * C + c) * 2] = s;
        part[((size_t)blockIdx.y * C + c) * 2 + 1] = qq;
    }
}

__global__ __launch_bounds__(256) void col_stats_final_kernel(int N, int C, int R, const double* __restrict__ part, float* __restrict__ mean,
                                                              float* __restrict__ rstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, qq = 0.0;
    for (int r = 0; r < R; ++r) { s += part[((size_t)r * C + c) * 2]; qq += part[((size_t)r * C + c) * 2 + 1]; }
    const double m = s / N;
    double var = qq / N - m * m;                                    // biased variance (InstanceNorm1d)
    if (var < 0) var = 0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + 1e-5));
}

// out = act( (a - mean_a) rstd_a + [ b normalised with (mean_b, rstd_b) if given, else b as is ] )
__global__ __launch_bounds__(256) void norm_apply_kernel(int N, int C, const float* __restrict__ a, int lda, const float* __restrict__ ma,
                                                         const float* __restrict__ ra, const float* __restrict__ b, int ldb,
                                                         const float* __restrict__ mb, const float* __restrict__ rb, float slope, int act,
                                                         float* __restrict__ out, int ldo) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * C) return;
    const int r = (int)(e / C), c = (int)(e % C);
    float v = (a[(size_t)r * lda + c] - ma[c]) * ra[c];
    if (b) {
        float u = b[(size_t)r * ldb + c];
        if (mb) u = (u - mb[c]) * rb[c];
        v += u;
    }
    if (act) v = v > 0.f ? v : v * slope;
    out[(size_t)r * ldo + c] = v;
}

size_t col_stats_workspace_bytes(int N, int C) {
    const int R = (N + 255) / 256 > 64 ? 64 : (N + 255) / 256;
    return (size_t)(R > 0 ? R : 1) * C * 2 * sizeof(double);
}

int launch_col_stats(int N, int C, const float* x, int ldx, float* mean, float* rstd, void* ws, size_t ws_bytes, hipStream_t st) {
    if (N <= 0 || C <= 0) return DR_OK;
    if (!ws || ws_bytes < col_stats_workspace_bytes(N, C)) return DR_EWORKSPACE;
    int R = (N + 255) / 256;
    if (R > 64) R = 64;
    const int rows_per = (N + R - 1) / R;
    hipLaunchKernelGGL(col_stats_partial_kernel, dim3((C + 31) / 32, R), dim3(256), 0, st, N, C, x, ldx, rows_per, (double*)ws);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_stats_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, N, C, R, (const double*)ws, mean, rstd);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_norm_apply(int N, int C, const float* a, int lda, const float* ma, const float* ra, const float* b, int ldb, const float* mb,
                      const float* rb, float slope, int act, float* out, int ldo, hipStream_t st) {
    if (N <= 0 || C <= 0) return DR_OK;
    const size_t n = (size_t)N * C;
    hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, N, C, a, lda, ma, ra, b, ldb, mb, rb, slope, act,
                       out, ldo);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---- max_pool / closest_pool (blocks.py:56-87): out[i][c] = max_h x~[inds[i][h]][c] (x~ = x with a zero shadow row), or the row
// of the FIRST index only (mode 1)
__global__ __launch_bounds__(256) void gather_pool_kernel(int n2, int H, int ldi, int d, const float* __restrict__ x, int n1,
                                                          const long long* __restrict__ inds, int first_only, float* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n2 * d) return;
    const int i = (int)(e / d), c = (int)(e % d);
    float m = 0.f;
    if (first_only) {
        const long long id = inds[(size_t)i * ldi];
        m = (id >= 0 && id < n1) ? x[(size_t)id * d + c] : 0.f;
    } else {
        m = -INFINITY;
        for (int h = 0; h < H; ++h) {
            const long long id = inds[(size_t)i * ldi + h];
            const float v = (id >= 0 && id < n1) ? x[(size_t)id * d + c] : 0.f;
            m = fmaxf(m, v);
        }
    }
    out[e] = m;
}

int launch_gather_pool(int n2, int H, int ldi, int d, const float* x, int n1, const long long* inds, int first_only, float* out, hipStream_t st) {
    if (n2 <= 0 || d <= 0) return DR_OK;
    const size_t n = (size_t)n2 * d;
    hipLaunchKernelGGL(gather_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n2, H, ldi, d, x, n1, inds, first_only, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" {

int dr_kpconv_gather_mode_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                              const float* x, const float* kernel_points, float extent, int influence, int closest, float* weighted,
                              int ld_weighted, void* stream) {
    if (Nq < 0 || Ns < 1 || !q_pts || !s_pts || !neighb_inds || !x || !kernel_points || !weighted || extent <= 0.f || influence < 0 || influence > 2)
        return DR_EINVAL;
    return launch_kpconv_gather(Nq, Ns, H, Cin, K, q_pts, s_pts, (const long long*)neighb_inds, x, kernel_points, extent, weighted,
                                ld_weighted, (hipStream_t)stream, influence | (closest ? KP_CLOSEST : 0));
}
int dr_kpconv_gather_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                         const float* x, const float* kernel_points, float extent, float* weighted, int ld_weighted, void* stream) {
    return dr_kpconv_gather_mode_f32(Nq, Ns, H, Cin, K, q_pts, s_pts, neighb_inds, x, kernel_points, extent, DR_KP_LINEAR, 0, weighted, ld_weighted, stream);
}

size_t dr_col_stats_workspace_bytes(int N, int C) { return (N > 0 && C > 0) ? col_stats_workspace_bytes(N, C) : 0; }

int dr_col_stats_f32(int N, int C, const float* x, int ldx, float* mean, float* rstd, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (N < 1 || C < 1 || !x || !mean || !rstd || ldx < C) return DR_EINVAL;
    return launch_col_stats(N, C, x, ldx, mean, rstd, workspace, workspace_bytes, (hipStream_t)stream);
}

int dr_norm_apply_f32(int N, int C, const float* a, int lda, const float* mean_a, const float* rstd_a, const float* b, int ldb,
                      const float* mean_b, const float* rstd_b, float leaky_slope, int activate, float* out, int ldo, void* stream) {
    if (N < 0 || C < 1 || !a || !mean_a || !rstd_a || !out || (mean_b && !rstd_b) || (mean_b && !b)) return DR_EINVAL;
    return launch_norm_apply(N, C, a, lda, mean_a, rstd_a, b, ldb, mean_b, rstd_b, leaky_slope, activate, out, ldo, (hipStream_t)stream);
}

int dr_gather_pool_f32(int n2, int H, int ld_inds, int d, const float* x, int n1, const int64_t* inds, int first_only, float* out,
                       void* stream) {
    if (n2 < 0 || d < 1 || H < 1 || ld_inds < (first_only ? 1 : H) || !x || !inds || !out || n1 < 0) return DR_EINVAL;
    return launch_gather_pool(n2, H, ld_inds, d, x, n1, (const long long*)inds, first_only, out, (hipStream_t)stream);
}

}  // extern "C"
