// backbone_bwd.hip -- backward kernels of the KPFCN backbone's coarse phase (second half; 3D/models/blocks.py).
//   kpconv_gather_backward   d loss / d x of KPConv's gather / influence / neighbour reduction (blocks.py:288-393): the influences
//                            depend on points only (no gradient), the neighbour count of the normalisation is piecewise constant;
//                            a neighbour's feature row receives sum_k w[q][k][h] g[q][k Cin + c] / num_q from every query it serves
//                            (fp32 atomics: the neighbour relation of a pooled level is not symmetric, so there is no gather form)
//   norm_backward            BatchNormBlock (InstanceNorm1d over the points, per channel, no affine; blocks.py:430-446) fused with
//                            LeakyReLU and the residual sum, as norm_apply computes it: two column reductions (sum g, sum g xhat)
//                            in float64 over a fixed grid of partials, then the element-wise part
//   gather_pool_backward     max_pool / closest_pool (blocks.py:56-87): the gradient goes to the FIRST maximal neighbour (torch.max)
// The nn.Linear halves (KPConv's single GEMM, the unary blocks, coarse_out) are products on the library's GEMM (autograd wrappers).
#include "kernels.h"

namespace dr {

constexpr int KPB_MAXK = 16, KPB_MAXH = 64;

template <int CPL>
__global__ __launch_bounds__(256) void kpconv_gather_backward_kernel(int Nq, int Ns, int H, int Cin, int K, const float* __restrict__ q_pts,
                                                                     const float* __restrict__ s_pts, const long long* __restrict__ nb,
                                                                     const float* __restrict__ x, const float* __restrict__ kp, float extent, int mode,
                                                                     const float* __restrict__ gwf, int ldw, float* __restrict__ gx) {
    __shared__ __attribute__((aligned(16))) float s_w[4][KPB_MAXH * KPB_MAXK];
    __shared__ int s_idx[4][KPB_MAXH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= Nq) return;
    const float qx = q_pts[q * 3], qy = q_pts[q * 3 + 1], qz = q_pts[q * 3 + 2];
    if (lane < H) {
        const long long id = nb[(size_t)q * H + lane];
        const bool shadow = id >= Ns || id < 0;
        s_idx[w][lane] = shadow ? -1 : (int)id;
        const float nx = (shadow ? 1e6f : s_pts[id * 3]) - qx, ny = (shadow ? 1e6f : s_pts[id * 3 + 1]) - qy,
                    nz = (shadow ? 1e6f : s_pts[id * 3 + 2]) - qz;
        float d2[KPB_MAXK];
        int kmin = 0;                                               // the neighbour's nearest kernel point (first minimum, as torch.argmin)
#pragma unroll
        for (int k = 0; k < KPB_MAXK; ++k) {
            d2[k] = INFINITY;
            if (k < K) {
                const float dx = nx - kp[k * 3], dy = ny - kp[k * 3 + 1], dz = nz - kp[k * 3 + 2];
                d2[k] = dx * dx + dy * dy + dz * dz;
                if (d2[k] < d2[kmin]) kmin = k;
            }
        }
#pragma unroll
        for (int k = 0; k < KPB_MAXK; ++k) {
            float wv = k < K ? kp_influence(mode, d2[k], extent) : 0.f;         // 'linear' in every shipped configuration (blocks.py:309-312)
            if ((mode & KP_CLOSEST) && k != kmin) wv = 0.f;
            s_w[w][lane * KPB_MAXK + k] = wv;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the forward's neighbour count: neighbours with a positive feature sum (same summation order as kpconv_gather_kernel)
    int num = 0;
    for (int h = 0; h < H; ++h) {
        const int id = s_idx[w][h];
        if (id < 0) continue;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            part += ch < Cin ? x[(size_t)id * Cin + ch] : 0.f;
        }
        part = wave_sum(part);
        num += part > 0.f ? 1 : 0;
    }
    const float inv = 1.f / (float)(num > 1 ? num : 1);
    // upstream gradient of this query's K x Cin block, the lane's channels
    float gw[CPL][KPB_MAXK];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = lane + 64 * c;
#pragma unroll
        for (int k = 0; k < KPB_MAXK; ++k) gw[c][k] = (k < K && ch < Cin) ? gwf[(size_t)q * ldw + k * Cin + ch] * inv : 0.f;
    }
    for (int h = 0; h < H; ++h) {
        const int id = s_idx[w][h];
        if (id < 0) continue;
        float wk[KPB_MAXK];
#pragma unroll
        for (int k4 = 0; k4 < KPB_MAXK / 4; ++k4) {
            const float4 t4 = *reinterpret_cast<const float4*>(&s_w[w][h * KPB_MAXK + 4 * k4]);
            wk[4 * k4] = t4.x; wk[4 * k4 + 1] = t4.y; wk[4 * k4 + 2] = t4.z; wk[4 * k4 + 3] = t4.w;
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            float g = 0.f;
#pragma unroll
            for (int k = 0; k < KPB_MAXK; ++k) g = fmaf(wk[k], gw[c][k], g);
            if (ch < Cin && g != 0.f) atomicAdd(gx + (size_t)id * Cin + ch, g);
        }
    }
}

int launch_kpconv_gather_backward(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const long long* nb,
                                  const float* x, const float* kp, float extent, const float* gwf, int ldw, float* gx, hipStream_t st, int mode) {
    if (Ns > 0) DR_HIP_CHECK(hipMemsetAsync(gx, 0, (size_t)Ns * Cin * sizeof(float), st));
    if (Nq <= 0) return DR_OK;
    if (K > KPB_MAXK || H > KPB_MAXH || H < 1 || Cin < 1 || Cin > 512 || ldw < K * Cin) return DR_ENOSUP;
    const dim3 grid((Nq + 3) / 4), blk(256);
#define KPB_LAUNCH(C_) hipLaunchKernelGGL(kpconv_gather_backward_kernel<C_>, grid, blk, 0, st, Nq, Ns, H, Cin, K, q_pts, s_pts, nb, x, kp, extent, mode, gwf, ldw, gx)
    if (Cin <= 64) KPB_LAUNCH(1);
    else if (Cin <= 128) KPB_LAUNCH(2);
    else if (Cin <= 256) KPB_LAUNCH(4);
    else KPB_LAUNCH(8);
#undef KPB_LAUNCH
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---- norm_apply backward ------------------------------------------------------------------------------------------------------------
// forward: out = act( xhat_a + u ),  xhat_a = (a - ma) ra,  u = (b - mb) rb | b | 0.   g = g_out act'(out)  (LeakyReLU: the sign of out).
// stage 1 partials per column over a fixed grid of row slabs: [sum g, sum g xhat_a, sum g xhat_b] in float64
__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(int N, int C, const float* __restrict__ gout, int ldg, const float* __restrict__ out,
                                                               int ldo, const float* __restrict__ a, int lda, const float* __restrict__ ma,
                                                               const float* __restrict__ ra, const float* __restrict__ b, int ldb,
                                                               const float* __restrict__ mb, const float* __restrict__ rb, float slope, int act,
                                                               int rows_per, double* __restrict__ part) {
    __shared__ double s0[8][32], s1[8][32], s2[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rows_per, r1 = min(N, r0 + rows_per);
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (c < C) {
        const float m_a = ma[c], r_a = ra[c], m_b = mb ? mb[c] : 0.f, r_b = mb ? rb[c] : 0.f;
        for (int r = r0 + rl; r < r1; r += 8) {
            float g = gout[(size_t)r * ldg + c];
            if (act && !(out[(size_t)r * ldo + c] > 0.f)) g *= slope;
            t0 += (double)g;
            t1 += (double)g * (double)((a[(size_t)r * lda + c] - m_a) * r_a);
            if (mb) t2 += (double)g * (double)((b[(size_t)r * ldb + c] - m_b) * r_b);
        }
    }
    s0[rl][cl] = t0; s1[rl][cl] = t1; s2[rl][cl] = t2;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < 8; ++k) { t0 += s0[k][cl]; t1 += s1[k][cl]; t2 += s2[k][cl]; }
        double* p = part + ((size_t)blockIdx.y * C + c) * 3;
        p[0] = t0; p[1] = t1; p[2] = t2;
    }
}
__global__ __launch_bounds__(256) void norm_bwd_final_kernel(int N, int C, int R, const double* __restrict__ part, float* __restrict__ red) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < R; ++r) { const double* p = part + ((size_t)r * C + c) * 3; t0 += p[0]; t1 += p[1]; t2 += p[2]; }
    red[c] = (float)(t0 / N); red[C + c] = (float)(t1 / N); red[2 * C + c] = (float)(t2 / N);       // column means
}
// ga = ra (g - mean g - xhat_a mean(g xhat_a));  gb = rb (g - mean g - xhat_b mean(g xhat_b)) | g
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(int N, int C, const float* __restrict__ gout, int ldg, const float* __restrict__ out,
                                                             int ldo, const float* __restrict__ a, int lda, const float* __restrict__ ma,
                                                             const float* __restrict__ ra, const float* __restrict__ b, int ldb,
                                                             const float* __restrict__ mb, const float* __restrict__ rb, float slope, int act,
                                                             const float* __restrict__ red, float* __restrict__ ga, int ldga,
                                                             float* __restrict__ gb, int ldgb) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * C) return;
    const int r = (int)(e / C), c = (int)(e % C);
    float g = gout[(size_t)r * ldg + c];
    if (act && !(out[(size_t)r * ldo + c] > 0.f)) g *= slope;
    const float xa = (a[(size_t)r * lda + c] - ma[c]) * ra[c];
    ga[(size_t)r * ldga + c] = ra[c] * (g - red[c] - xa * red[C + c]);
    if (gb) {
        if (mb) {
            const float xb = (b[(size_t)r * ldb + c] - mb[c]) * rb[c];
            gb[(size_t)r * ldgb + c] = rb[c] * (g - red[c] - xb * red[2 * C + c]);
        } else gb[(size_t)r * ldgb + c] = g;
    }
}

size_t norm_backward_workspace_bytes(int N, int C) {
    const int R = (N + 255) / 256 > 64 ? 64 : (N + 255) / 256;
    return (size_t)(R > 0 ? R : 1) * C * 3 * sizeof(double) + (size_t)3 * C * sizeof(float) + 64;
}

int launch_norm_backward(int N, int C, const float* gout, int ldg, const float* out, int ldo, const float* a, int lda, const float* ma,
                         const float* ra, const float* b, int ldb, const float* mb, const float* rb, float slope, int act, float* ga, int ldga,
                         float* gb, int ldgb, void* ws, size_t ws_bytes, hipStream_t st) {
    if (N <= 0 || C <= 0) return DR_OK;
    if (!ws || ws_bytes < norm_backward_workspace_bytes(N, C)) return DR_EWORKSPACE;
    int R = (N + 255) / 256;
    if (R > 64) R = 64;
    const int rows_per = (N + R - 1) / R;
    double* part = (double*)ws;
    float* red = (float*)((char*)ws + (size_t)R * C * 3 * sizeof(double));
    hipLaunchKernelGGL(norm_bwd_partial_kernel, dim3((C + 31) / 32, R), dim3(256), 0, st, N, C, gout, ldg, out, ldo, a, lda, ma, ra, b, ldb, mb, rb,
                       slope, act, rows_per, part);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_bwd_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, N, C, R, (const double*)part, red);
    DR_LAUNCH_CHECK();
    const size_t n = (size_t)N * C;
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, N, C, gout, ldg, out, ldo, a, lda, ma, ra, b, ldb,
                       mb, rb, slope, act, (const float*)red, ga, ldga, gb, ldgb);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---- max_pool / closest_pool backward -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_pool_backward_kernel(int n2, int H, int ldi, int d, const float* __restrict__ x, int n1,
                                                                   const long long* __restrict__ inds, int first_only,
                                                                   const float* __restrict__ gout, float* __restrict__ gx) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n2 * d) return;
    const int i = (int)(e / d), c = (int)(e % d);
    long long tgt = -1;
    if (first_only) {
        tgt = inds[(size_t)i * ldi];
    } else {
        float m = -INFINITY;
        for (int h = 0; h < H; ++h) {
            const long long id = inds[(size_t)i * ldi + h];
            const float v = (id >= 0 && id < n1) ? x[(size_t)id * d + c] : 0.f;
            if (v > m) { m = v; tgt = id; }                       // first maximal neighbour (torch.max over dim 1)
        }
    }
    if (tgt >= 0 && tgt < n1) atomicAdd(gx + (size_t)tgt * d + c, gout[e]);
}

int launch_gather_pool_backward(int n2, int H, int ldi, int d, const float* x, int n1, const long long* inds, int first_only, const float* gout,
                                float* gx, hipStream_t st) {
    if (n1 > 0 && d > 0) DR_HIP_CHECK(hipMemsetAsync(gx, 0, (size_t)n1 * d * sizeof(float), st));
    if (n2 <= 0 || d <= 0) return DR_OK;
    const size_t n = (size_t)n2 * d;
    hipLaunchKernelGGL(gather_pool_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n2, H, ldi, d, x, n1, inds, first_only, gout, gx);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace dr

using namespace dr;

extern "C" {

int dr_kpconv_gather_backward_mode_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                                       const float* x, const float* kernel_points, float extent, int influence, int closest,
                                       const float* grad_weighted, int ld_weighted, float* grad_x, void* stream) {
    if (Nq < 0 || Ns < 1 || !q_pts || !s_pts || !neighb_inds || !x || !kernel_points || !grad_weighted || !grad_x || extent <= 0.f || influence < 0 ||
        influence > 2) return DR_EINVAL;
    return launch_kpconv_gather_backward(Nq, Ns, H, Cin, K, q_pts, s_pts, (const long long*)neighb_inds, x, kernel_points, extent, grad_weighted,
                                         ld_weighted, grad_x, (hipStream_t)stream, influence | (closest ? KP_CLOSEST : 0));
}
int dr_kpconv_gather_backward_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                                  const float* x, const float* kernel_points, float extent, const float* grad_weighted, int ld_weighted,
                                  float* grad_x, void* stream) {
    return dr_kpconv_gather_backward_mode_f32(Nq, Ns, H, Cin, K, q_pts, s_pts, neighb_inds, x, kernel_points, extent, DR_KP_LINEAR, 0, grad_weighted,
                                              ld_weighted, grad_x, stream);
}

size_t dr_norm_backward_workspace_bytes(int N, int C) { return (N > 0 && C > 0) ? norm_backward_workspace_bytes(N, C) : 0; }

int dr_norm_backward_f32(int N, int C, const float* grad_out, int ldg, const float* out, int ldo, const float* a, int lda, const float* mean_a,
                         const float* rstd_a, const float* b, int ldb, const float* mean_b, const float* rstd_b, float leaky_slope, int activate,
                         float* grad_a, int ldga, float* grad_b, int ldgb, void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 1 || C < 1 || !grad_out || !a || !mean_a || !rstd_a || !grad_a || (activate && !out) || (mean_b && (!rstd_b || !b)) || (grad_b && !b)) return DR_EINVAL;
    return launch_norm_backward(N, C, grad_out, ldg, out, ldo, a, lda, mean_a, rstd_a, b, ldb, mean_b, rstd_b, leaky_slope, activate, grad_a, ldga,
                                grad_b, ldgb, workspace, workspace_bytes, (hipStream_t)stream);
}

int dr_gather_pool_backward_f32(int n2, int H, int ld_inds, int d, const float* x, int n1, const int64_t* inds, int first_only, const float* grad_out,
                                float* grad_x, void* stream) {
    if (n2 < 0 || d < 1 || H < 1 || ld_inds < (first_only ? 1 : H) || !x || !inds || !grad_out || !grad_x || n1 < 0) return DR_EINVAL;
    return launch_gather_pool_backward(n2, H, ld_inds, d, x, n1, (const long long*)inds, first_only, grad_out, grad_x, (hipStream_t)stream);
}

}  // extern "C"
