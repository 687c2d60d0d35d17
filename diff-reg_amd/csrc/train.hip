// train.hip -- forward half of the training branch around the loop's kernels (HBM-bound, one pass each):
//   match_matrix   match_2_conf_matrix / matrix_gt                  (3D/models/loss.py:316-320, pipeline.py:203-206)
//   gt_noising     structured noise + q_sample + nan_to_num + min   (pipeline.py:209-214, q_sample :84-95)
//   focal_loss     compute_correspondence_loss                      (loss.py:273-314)
//   match_recall   compute_match_recall                             (loss.py:323-345)
//   motion_l1      L1 motion term of ge_coarse_loss                 (loss.py:108-128)
// Reductions are two kernels (fixed grid of partials, one finishing workgroup that adds them in index order): results do
// not depend on the launch's scheduling.
#include "kernels.h"

namespace dr {
namespace {

constexpr int TR_BLOCKS = 1024;      // partials of a reduction
constexpr int TR_THREADS = 256;

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* s) {      // s: [TR_THREADS / 64]
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) s[wave_id()] = v;
    __syncthreads();
    T r = s[0];
    for (int k = 1; k < TR_THREADS / 64; ++k) r += s[k];
    return r;
}

// -------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void match_matrix_kernel(int K, long long NM, int N, int M, int P, const long long* __restrict__ m,
                                                           float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const long long b = m[3 * k], i = m[3 * k + 1], j = m[3 * k + 2];
    if (b < 0 || b >= P || i < 0 || i >= N || j < 0 || j >= M) return;      // (the reference would raise)
    out[b * NM + i * M + j] = 1.0f;
}

// -------------------------------------------------------------------------------------------------------------------------------
// noise = ((|r| % 1) * (|r| / r)) * 1.5 in float32 (r = 0 gives 0 * nan = nan -> 0 by nan_to_num); x = sa * gt + sb * noise in
// float64 (a float64 [1,1,1] tensor times a float32 matrix promotes; separate multiply and add, no contraction)
__device__ __forceinline__ double noised_value(float g, float r, double sa, double sb) {
    const float a = fabsf(r);
    const float frac = a - floorf(a);                     // fmod(a, 1) for a >= 0: exact
    const float sgn = __fdiv_rn(a, r);                    // +-1, nan at r = 0
    const float n = __fmul_rn(__fmul_rn(frac, sgn), 1.5f);
    double x = __dadd_rn(__dmul_rn(sa, (double)g), __dmul_rn(sb, (double)n));
    if (x != x) x = 0.0;                                  // nan_to_num(nan = 0); +-inf cannot occur (|noise| < 1.5)
    return x;
}

__global__ __launch_bounds__(TR_THREADS) void gt_noising_min_kernel(long long n, const float* __restrict__ gt, const float* __restrict__ r,
                                                                    double sa, double sb, double* __restrict__ out, double* __restrict__ part) {
    __shared__ double s[TR_THREADS / 64];
    double m = INFINITY;
    for (long long e = (long long)blockIdx.x * TR_THREADS + threadIdx.x; e < n; e += (long long)TR_BLOCKS * TR_THREADS) {
        const double x = noised_value(gt[e], r[e], sa, sb);
        out[e] = x;
        m = fmin(m, x);
    }
    m = wave_min(m);
    if (lane_id() == 0) s[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < TR_THREADS / 64; ++k) m = fmin(m, s[k]);
        part[blockIdx.x] = fmin(m, s[0]);
    }
}

__global__ __launch_bounds__(TR_THREADS) void gt_noising_shift_kernel(long long n, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double s[TR_THREADS / 64];
    double m = INFINITY;
    for (int k = threadIdx.x; k < TR_BLOCKS; k += TR_THREADS) m = fmin(m, part[k]);
    m = wave_min(m);
    if (lane_id() == 0) s[wave_id()] = m;
    __syncthreads();
    m = s[0];
    for (int k = 1; k < TR_THREADS / 64; ++k) m = fmin(m, s[k]);
    for (long long e = (long long)blockIdx.x * TR_THREADS + threadIdx.x; e < n; e += (long long)TR_BLOCKS * TR_THREADS) out[e] = out[e] - m;
}

// -------------------------------------------------------------------------------------------------------------------------------
// focal terms in float32 in the reference's operation order ((-alpha * pow) * log), summed in float64
struct FocalArgs {
    const float* conf; const float* gt; const float* weight; long long n; float alpha, gamma, pos_w, neg_w; int dual_softmax;
    double* part;      // [TR_BLOCKS][4]: sum pos, n pos, sum neg, n neg
    float* loss;
};

__device__ __forceinline__ float powg(float x, float g) { return g == 2.0f ? x * x : (g == 1.0f ? x : powf(x, g)); }

__global__ __launch_bounds__(TR_THREADS) void focal_partial_kernel(FocalArgs A) {
    __shared__ double s[TR_THREADS / 64];
    double sp = 0, np = 0, sn = 0, nn = 0;
    for (long long e = (long long)blockIdx.x * TR_THREADS + threadIdx.x; e < A.n; e += (long long)TR_BLOCKS * TR_THREADS) {
        const float g = A.gt[e];
        const float c = fminf(fmaxf(A.conf[e], 1e-6f), 1.0f - 1e-6f);
        if (g == 1.0f) {
            float l = __fmul_rn(__fmul_rn(-A.alpha, powg(1.0f - c, A.gamma)), logf(c));
            if (A.dual_softmax && A.weight) l = __fmul_rn(l, A.weight[e]);
            sp += (double)l; np += 1.0;
        } else if (g == 0.0f && !A.dual_softmax) {
            const float l = __fmul_rn(__fmul_rn(-A.alpha, powg(c, A.gamma)), logf(1.0f - c));
            sn += (double)l; nn += 1.0;
        } else if (g == 0.0f) {
            nn += 1.0;
        }
    }
    sp = block_sum(sp, s); np = block_sum(np, s); sn = block_sum(sn, s); nn = block_sum(nn, s);
    if (threadIdx.x == 0) {
        double* p = A.part + 4 * blockIdx.x;
        p[0] = sp; p[1] = np; p[2] = sn; p[3] = nn;
    }
}

__global__ __launch_bounds__(64) void focal_final_kernel(FocalArgs A) {
    if (threadIdx.x != 0) return;
    double sp = 0, np = 0, sn = 0, nn = 0;
    for (int k = 0; k < TR_BLOCKS; ++k) { sp += A.part[4 * k]; np += A.part[4 * k + 1]; sn += A.part[4 * k + 2]; nn += A.part[4 * k + 3]; }
    // corner cases (loss.py:287-297): no positive (negative) entry -> entry [0,0,0] stands in with weight 0: the term is 0 x finite.
    // dual_softmax without any negative entry: weight[0,0,0] = 0 (:294) zeroes the weighted term of entry 0, which IS positive
    if (A.dual_softmax && A.weight && nn == 0 && A.n > 0 && A.gt[0] == 1.0f) {
        const float c = fminf(fmaxf(A.conf[0], 1e-6f), 1.0f - 1e-6f);
        sp -= (double)__fmul_rn(__fmul_rn(__fmul_rn(-A.alpha, powg(1.0f - c, A.gamma)), logf(c)), A.weight[0]);
    }
    const float mp = np > 0 ? (float)(sp / np) : 0.0f;
    const float mn = nn > 0 ? (float)(sn / nn) : 0.0f;
    const float pw = np > 0 ? A.pos_w : 0.0f, nw = nn > 0 ? A.neg_w : 0.0f;
    *A.loss = A.dual_softmax ? __fmul_rn(pw, mp) : __fadd_rn(__fmul_rn(pw, mp), __fmul_rn(nw, mn));
}

// -------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mark_pred_kernel(int K, long long NM, int N, int M, int P, const long long* __restrict__ m,
                                                        uint8_t* __restrict__ mark) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const long long b = m[3 * k], i = m[3 * k + 1], j = m[3 * k + 2];
    if (b < 0 || b >= P || i < 0 || i >= N || j < 0 || j >= M) return;
    mark[b * NM + i * M + j] = 1;
}

__global__ __launch_bounds__(TR_THREADS) void recall_partial_kernel(long long n, const float* __restrict__ gt, const uint8_t* __restrict__ mark,
                                                                    double* __restrict__ part) {
    __shared__ double s[TR_THREADS / 64];
    double tp = 0, ng = 0;
    for (long long e = (long long)blockIdx.x * TR_THREADS + threadIdx.x; e < n; e += (long long)TR_BLOCKS * TR_THREADS) {
        const float g = gt[e];
        // true_positive = (pred == gt) * gt  (loss.py:339): gt where the two agree
        if ((mark[e] ? 1.0f : 0.0f) == g) tp += (double)g;
        ng += (double)g;
    }
    tp = block_sum(tp, s); ng = block_sum(ng, s);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = tp; part[2 * blockIdx.x + 1] = ng; }
}

__global__ __launch_bounds__(64) void recall_final_kernel(const double* __restrict__ part, int K, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double tp = 0, ng = 0;
    for (int k = 0; k < TR_BLOCKS; ++k) { tp += part[2 * k]; ng += part[2 * k + 1]; }
    out[0] = (float)tp / (float)ng;                       // recall (nan without any ground-truth entry, like the reference)
    out[1] = (float)tp / (float)(K > 1 ? K : 1);          // precision
}

// -------------------------------------------------------------------------------------------------------------------------------
// e1[b][i] = sum_c |(R_p s + t_p - s) - (R_g (s + flow) + t_g - s)|_c over the rows inside the overlap mask; mean over those rows
struct MotionArgs {
    const float* s; const float* flow; const float* Rp; const float* tp; const float* Rg; const float* tg; const uint8_t* mask; int P, N;
    double* part;      // [TR_BLOCKS][2]
    float* loss;
};

__global__ __launch_bounds__(TR_THREADS) void motion_partial_kernel(MotionArgs A) {
    __shared__ double s[TR_THREADS / 64];
    double se = 0, cnt = 0;
    const long long n = (long long)A.P * A.N;
    for (long long e = (long long)blockIdx.x * TR_THREADS + threadIdx.x; e < n; e += (long long)TR_BLOCKS * TR_THREADS) {
        if (!A.mask[e]) continue;
        const int b = (int)(e / A.N);
        const float x = A.s[3 * e], y = A.s[3 * e + 1], z = A.s[3 * e + 2];
        float dx = x, dy = y, dz = z;
        if (A.flow) { dx += A.flow[3 * e]; dy += A.flow[3 * e + 1]; dz += A.flow[3 * e + 2]; }
        const float* Rp = A.Rp + 9 * b; const float* Rg = A.Rg + 9 * b;
        float e1 = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float wp = (Rp[3 * c] * x + Rp[3 * c + 1] * y + Rp[3 * c + 2] * z) + A.tp[3 * b + c];
            const float wg = (Rg[3 * c] * dx + Rg[3 * c + 1] * dy + Rg[3 * c + 2] * dz) + A.tg[3 * b + c];
            const float sc = c == 0 ? x : (c == 1 ? y : z);
            e1 += fabsf((wp - sc) - (wg - sc));
        }
        se += (double)e1; cnt += 1.0;
    }
    se = block_sum(se, s); cnt = block_sum(cnt, s);
    if (threadIdx.x == 0) { A.part[2 * blockIdx.x] = se; A.part[2 * blockIdx.x + 1] = cnt; }
}

__global__ __launch_bounds__(64) void motion_final_kernel(MotionArgs A) {
    if (threadIdx.x != 0) return;
    double se = 0, cnt = 0;
    for (int k = 0; k < TR_BLOCKS; ++k) { se += A.part[2 * k]; cnt += A.part[2 * k + 1]; }
    *A.loss = (float)(se / cnt);                          // mean of an empty selection is nan in the reference too
}

}  // namespace
}  // namespace dr

extern "C" {

size_t dr_train_workspace_bytes(int P, int N, int M) {
    if (P < 0 || N < 0 || M < 0) return 0;
    return (size_t)dr::TR_BLOCKS * 4 * sizeof(double) + (((size_t)P * N * M + 255) & ~(size_t)255);
}

int dr_match_matrix_f32(int P, int N, int M, int K, const int64_t* matches, float* out, void* stream) {
    if (P < 0 || N < 1 || M < 1 || K < 0 || !out || (K > 0 && !matches)) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DR_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)P * N * M * sizeof(float), st));
    if (K == 0 || P == 0) return DR_OK;
    hipLaunchKernelGGL(dr::match_matrix_kernel, dim3((K + 255) / 256), dim3(256), 0, st, K, (long long)N * M, N, M, P, (const long long*)matches, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_gt_noising_f64(int P, int N, int M, const float* matrix_gt, const float* randn, double sqrt_ac, double sqrt_one_minus_ac, double* out,
                      void* workspace, void* stream) {
    if (P < 1 || N < 1 || M < 1 || !matrix_gt || !randn || !out || !workspace) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)P * N * M;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(dr::gt_noising_min_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, n, matrix_gt, randn, sqrt_ac, sqrt_one_minus_ac, out, part);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::gt_noising_shift_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, n, part, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_focal_loss_f32(int P, int N, int M, const float* conf, const float* conf_gt, const float* weight, float alpha, float gamma, float pos_w,
                      float neg_w, int match_type, float* loss, void* workspace, void* stream) {
    if (P < 1 || N < 1 || M < 1 || !conf || !conf_gt || !loss || !workspace || (match_type != DR_MATCH_SINKHORN && match_type != DR_MATCH_DUAL_SOFTMAX))
        return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dr::FocalArgs A{conf, conf_gt, weight, (long long)P * N * M, alpha, gamma, pos_w, neg_w, match_type == DR_MATCH_DUAL_SOFTMAX, (double*)workspace, loss};
    hipLaunchKernelGGL(dr::focal_partial_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, A);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::focal_final_kernel, dim3(1), dim3(64), 0, st, A);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_match_recall_f32(int P, int N, int M, const float* conf_gt, int K, const int64_t* match_pred, float* recall_precision, void* workspace,
                        void* stream) {
    if (P < 1 || N < 1 || M < 1 || K < 0 || !conf_gt || !recall_precision || !workspace || (K > 0 && !match_pred)) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    uint8_t* mark = (uint8_t*)workspace + (size_t)dr::TR_BLOCKS * 4 * sizeof(double);
    const long long n = (long long)P * N * M;
    DR_HIP_CHECK(hipMemsetAsync(mark, 0, (size_t)n, st));
    if (K > 0) {
        hipLaunchKernelGGL(dr::mark_pred_kernel, dim3((K + 255) / 256), dim3(256), 0, st, K, (long long)N * M, N, M, P, (const long long*)match_pred, mark);
        DR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(dr::recall_partial_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, n, conf_gt, mark, part);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::recall_final_kernel, dim3(1), dim3(64), 0, st, part, K, recall_precision);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_motion_l1_f32(int P, int N, const float* s_pcd, const float* flow, const float* R_pred, const float* t_pred, const float* R_gt,
                     const float* t_gt, const uint8_t* overlap_mask, float* loss, void* workspace, void* stream) {
    if (P < 1 || N < 1 || !s_pcd || !R_pred || !t_pred || !R_gt || !t_gt || !overlap_mask || !loss || !workspace) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dr::MotionArgs A{s_pcd, flow, R_pred, t_pred, R_gt, t_gt, overlap_mask, P, N, (double*)workspace, loss};
    hipLaunchKernelGGL(dr::motion_partial_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, A);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::motion_final_kernel, dim3(1), dim3(64), 0, st, A);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // extern "C"

// ===============================================================================================================================
// Backward of the matching head's loss (first backward kernels of row f3): d loss / d conf of the focal loss and the backward of
// log_optimal_transport + exp + slice (3D/models/matching.py:61-93, 207-216) -> d loss / d scores, d loss / d bin_score.
//
// Forward (per pair, extended matrix Z = [[scores, a], [a, a]], i <= N, j <= M):  u^0 = v^0 = 0,
//     u^t_i = log mu_i - LSE_j(Z_ij + v^{t-1}_j),   v^t_j = log nu_j - LSE_i(Z_ij + u^t_i),   t = 1 .. T,
//     conf_ij = exp(Z_ij + u^T_i + v^T_j - norm),  i < N, j < M.
// Backward with D_ij = conf_ij dL/dconf_ij and the transport plans Pv^t_ij = exp(Z_ij + u^t_i + v^t_j - log nu_j),
// Pu^t_ij = exp(Z_ij + v^{t-1}_j + u^t_i - log mu_i):
//     vb^T_j = sum_i D_ij,   ub^T_i = sum_j D_ij - sum_j vb^T_j Pv^T_ij,
//     vb^{t-1}_j = - sum_i ub^t_i Pu^t_ij,   ub^{t-1}_i = - sum_j vb^{t-1}_j Pv^{t-1}_ij,
//     dL/dZ_ij = D_ij - sum_t (vb^t_j Pv^t_ij + ub^t_i Pu^t_ij);   dL/da = sum over the dustbin row and column of dL/dZ.
// One workgroup per pair; every step is one sweep over the matrix: row quantities by one wave per row (lanes over the columns, DPP
// reduction), column quantities by one thread per column (rows in order, coalesced): no atomics, results do not depend on scheduling.
// The 2 T + 2 T vectors live in the workspace (L2-resident), in double (see sk_backward_kernel).  The library runs the multi-launch form below
// (skb_*_kernel); the single-workgroup kernel stays as its reference (diagnostics knob DR_SKB_ONE_WG).
// ===============================================================================================================================
namespace dr {
namespace {

struct SkBwdArgs {
    const float* scores; const uint8_t* sm; const uint8_t* tm; const float* alpha; const float* gconf; float* gscores; float* galpha; void* ws;
    int N, M, iters;
};

template <typename AT>
__device__ __forceinline__ AT zval(const float* __restrict__ Z, int i, int j, int N, int M, AT a) {
    return (i < N && j < M) ? (AT)Z[(size_t)i * M + j] : a;
}
// (AT = double: only the SUMS that cancel need the width -- Z + u + v, the running maxima, the adjoint sums.  The transcendentals see O(1)
// arguments (a difference formed in double, a sum in [1, n]) and are taken in float32: 1e-7 relative, at a tenth of the double routines' cost)
__device__ __forceinline__ float xexp(float x) { return expf(x); }
__device__ __forceinline__ double xexp(double x) { return (double)expf((float)x); }
__device__ __forceinline__ float xlog(float x) { return logf(x); }
__device__ __forceinline__ double xlog(double x) { return (double)logf((float)x); }

// AT: the type of the dual variables, of the plans' exponents and of every sum (inputs and results are float32).  The library runs AT = double:
// at logits in the thousands (a sharp head) Z + u + v cancels three numbers of that size -- in float32 the exponent carries an absolute error of
// their ulp (2.4e-4 at 3 000) and the gradient came out 2.7e-3 of its maximum from the float64 value (the same recurrences in torch float32:
// 2.8e-3; torch autograd through the reference's float32 code: 1e-3); with double sums the step is at the float32 rounding of its inputs.
template <typename AT>
__global__ __launch_bounds__(1024) void sk_backward_kernel(SkBwdArgs A) {
    const int N = A.N, M = A.M, T = A.iters, t = threadIdx.x, lane = t & 63, w = t >> 6, NW = 16;
    const int pair = blockIdx.x;
    const float* __restrict__ Z = A.scores + (size_t)pair * N * M;
    const float* __restrict__ G = A.gconf + (size_t)pair * N * M;
    const AT a = (AT)*A.alpha;
    __shared__ int s_cnt[2];
    __shared__ AT s_red[16];
    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        for (int i = t; i < N; i += 1024) c0 += A.sm ? (A.sm[(size_t)pair * N + i] != 0) : 1;
        for (int j = t; j < M; j += 1024) c1 += A.tm ? (A.tm[(size_t)pair * M + j] != 0) : 1;
        if (c0) atomicAdd(&s_cnt[0], c0);
        if (c1) atomicAdd(&s_cnt[1], c1);
    }
    __syncthreads();
    const int ms = s_cnt[0], ns = s_cnt[1];
    // (the marginals are float32 numbers whatever the state's type: matching.py:69-70, 79-82 take .log() of int64 sums -- quirk Q22)
    const float normf = -logf((float)(ms + ns));
    const AT norm = (AT)normf, lmuN = (AT)(logf((float)ns) + normf), lnuM = (AT)(logf((float)ms) + normf);
    // workspace vectors of this pair: u[T][N+1], v[T+1][M+1] (v[0] = 0), ub[T][N+1], vb[T][M+1]
    const int R = N + 1, Cn = M + 1;
    AT* U = reinterpret_cast<AT*>(A.ws) + (size_t)pair * ((size_t)2 * T * R + (size_t)(2 * T + 1) * Cn);
    AT* V = U + (size_t)T * R;
    AT* UB = V + (size_t)(T + 1) * Cn;
    AT* VB = UB + (size_t)T * R;
    for (int j = t; j < Cn; j += 1024) V[j] = (AT)0;
    __syncthreads();
    auto lmu = [&](int i) -> AT { return i < N ? norm : lmuN; };
    auto lnu = [&](int j) -> AT { return j < M ? norm : lnuM; };

    // ---- forward, keeping every u^t, v^t
    for (int it = 1; it <= T; ++it) {
        const AT* vp = V + (size_t)(it - 1) * Cn;
        AT* un = U + (size_t)(it - 1) * R;
        for (int i = w; i < R; i += NW) {                        // u^t: one wave per row, online log-sum-exp over the columns
            AT mx = -INFINITY, s = 0;
            for (int j = lane; j < Cn; j += 64) {
                const AT x = zval<AT>(Z, i, j, N, M, a) + vp[j];
                if (x > mx) { s = s * xexp(mx - x) + (AT)1; mx = x; } else if (x > -INFINITY) s += xexp(x - mx);
            }
            const AT gm = wave_max(mx);
            s = (mx > -INFINITY) ? s * xexp(mx - gm) : (AT)0;
            s = wave_sum(s);
            if (lane == 0) un[i] = lmu(i) - (gm + xlog(s));
        }
        __threadfence_block();
        __syncthreads();
        AT* vn = V + (size_t)it * Cn;
        for (int j = t; j < Cn; j += 1024) {                     // v^t: one thread per column, rows in order
            AT mx = -INFINITY, s = 0;
            for (int i = 0; i < R; ++i) {
                const AT x = zval<AT>(Z, i, j, N, M, a) + un[i];
                if (x > mx) { s = s * xexp(mx - x) + (AT)1; mx = x; } else if (x > -INFINITY) s += xexp(x - mx);
            }
            vn[j] = lnu(j) - (mx + xlog(s));
        }
        __threadfence_block();
        __syncthreads();
    }
    // ---- backward vectors
    const AT* uT = U + (size_t)(T - 1) * R;
    const AT* vT = V + (size_t)T * Cn;
    {   // vb^T_j = sum_i D_ij  (D = 0 on the dustbin row / column)
        AT* vb = VB + (size_t)(T - 1) * Cn;
        for (int j = t; j < Cn; j += 1024) {
            AT s = 0;
            if (j < M)
                for (int i = 0; i < N; ++i) {
                    const AT z = (AT)Z[(size_t)i * M + j];
                    if (z > -INFINITY) s += xexp(z + uT[i] + vT[j] - norm) * (AT)G[(size_t)i * M + j];
                }
            vb[j] = s;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int it = T; it >= 1; --it) {
        const AT* u = U + (size_t)(it - 1) * R;
        const AT* v = V + (size_t)it * Cn;
        const AT* vprev = V + (size_t)(it - 1) * Cn;
        const AT* vb = VB + (size_t)(it - 1) * Cn;
        AT* ub = UB + (size_t)(it - 1) * R;
        for (int i = w; i < R; i += NW) {                        // ub^t_i = [t == T] sum_j D_ij - sum_j vb^t_j Pv^t_ij
            AT s = 0;
            for (int j = lane; j < Cn; j += 64) {
                const AT z = zval<AT>(Z, i, j, N, M, a);
                if (z > -INFINITY) {
                    s -= vb[j] * xexp(z + u[i] + v[j] - lnu(j));
                    if (it == T && i < N && j < M) s += xexp(z + u[i] + v[j] - norm) * (AT)G[(size_t)i * M + j];
                }
            }
            s = wave_sum(s);
            if (lane == 0) ub[i] = s;
        }
        __threadfence_block();
        __syncthreads();
        if (it > 1) {                                            // vb^{t-1}_j = - sum_i ub^t_i Pu^t_ij
            AT* vbp = VB + (size_t)(it - 2) * Cn;
            for (int j = t; j < Cn; j += 1024) {
                AT s = 0;
                for (int i = 0; i < R; ++i) {
                    const AT z = zval<AT>(Z, i, j, N, M, a);
                    if (z > -INFINITY) s -= ub[i] * xexp(z + vprev[j] + u[i] - lmu(i));
                }
                vbp[j] = s;
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    // ---- dL/dZ in one sweep; the dustbin entries go to dL/dalpha (wave partials in fixed order)
    AT ga = 0;
    for (int i = w; i < R; i += NW) {
        for (int j = lane; j < Cn; j += 64) {
            const AT z = zval<AT>(Z, i, j, N, M, a);
            AT g = 0;
            if (z > -INFINITY) {
                if (i < N && j < M) g = xexp(z + uT[i] + vT[j] - norm) * (AT)G[(size_t)i * M + j];
                for (int it = 1; it <= T; ++it) {
                    const AT ui = U[(size_t)(it - 1) * R + i];
                    g -= VB[(size_t)(it - 1) * Cn + j] * xexp(z + ui + V[(size_t)it * Cn + j] - lnu(j));
                    g -= UB[(size_t)(it - 1) * R + i] * xexp(z + V[(size_t)(it - 1) * Cn + j] + ui - lmu(i));
                }
            }
            if (i < N && j < M) A.gscores[((size_t)pair * N + i) * M + j] = (float)g;
            else ga += g;
        }
    }
    ga = wave_sum(ga);
    if (lane == 0) s_red[w] = ga;
    __syncthreads();
    if (t == 0) {
        AT s = 0;
        for (int k = 0; k < NW; ++k) s += s_red[k];
        A.galpha[pair] = s;
    }
}

// ---- the same recurrences as a SEQUENCE of launches (round 6): the single-workgroup kernel above is one CU working through ~4 T + 2 sweeps of the
// matrix -- 1.5 ms at 375 x 381 with double sums, the largest kernel of a training step.  Here every sweep is a launch over the whole chip: row
// quantities by one wave per row, column quantities by 64 columns x 16 row parts per workgroup (partials combined through LDS in a fixed order);
// 3 T + 4 launches of 5 - 10 us.  Same vectors in the workspace, same double sums, same fixed summation orders (bit-reproducible).
struct SkbCtx {
    const float* Z; const float* G; const uint8_t* sm; const uint8_t* tm; const float* alpha;
    double* ws; size_t ws_stride;                        // per pair: [hdr 8][U T R][V (T+1) Cn][UB T R][VB T Cn][GA R]
    float* gscores; float* galpha;
    int N, M, T;
};
__device__ __forceinline__ double* skb_pair(const SkbCtx& A, int pair) { return A.ws + (size_t)pair * A.ws_stride; }
struct SkbVec { double *U, *V, *UB, *VB, *GA; double norm, lmuN, lnuM, a; };
__device__ __forceinline__ SkbVec skb_vec(const SkbCtx& A, int pair) {
    double* w = skb_pair(A, pair);
    const size_t R = A.N + 1, Cn = A.M + 1, T = A.T;
    SkbVec v;
    v.norm = w[0]; v.lmuN = w[1]; v.lnuM = w[2]; v.a = w[3];
    v.U = w + 8; v.V = v.U + T * R; v.UB = v.V + (T + 1) * Cn; v.VB = v.UB + T * R; v.GA = v.VB + T * Cn;
    return v;
}
__device__ __forceinline__ double skb_z(const SkbCtx& A, int pair, int i, int j, double a) {
    return (i < A.N && j < A.M) ? (double)A.Z[((size_t)pair * A.N + i) * A.M + j] : a;
}

// header of a pair: the float32 marginals (quirk Q22) and alpha as doubles; v^0 = 0
__global__ __launch_bounds__(256) void skb_prep_kernel(SkbCtx A) {
    const int pair = blockIdx.x, t = threadIdx.x, N = A.N, M = A.M;
    __shared__ int s_cnt[2];
    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    int c0 = 0, c1 = 0;
    for (int i = t; i < N; i += 256) c0 += A.sm ? (A.sm[(size_t)pair * N + i] != 0) : 1;
    for (int j = t; j < M; j += 256) c1 += A.tm ? (A.tm[(size_t)pair * M + j] != 0) : 1;
    if (c0) atomicAdd(&s_cnt[0], c0);
    if (c1) atomicAdd(&s_cnt[1], c1);
    __syncthreads();
    double* w = skb_pair(A, pair);
    if (t == 0) {
        const int ms = s_cnt[0], ns = s_cnt[1];
        const float normf = -logf((float)(ms + ns));
        w[0] = (double)normf; w[1] = (double)(logf((float)ns) + normf); w[2] = (double)(logf((float)ms) + normf); w[3] = (double)*A.alpha;
    }
    double* V0 = w + 8 + (size_t)A.T * (N + 1);
    for (int j = t; j <= M; j += 256) V0[j] = 0.0;
}

// what a sweep computes
enum { SKB_U = 0, SKB_UB = 1, SKB_FINAL = 2, SKB_V = 3, SKB_VBT = 4, SKB_VB = 5 };

// row sweeps: one wave per row (4 rows per workgroup), lanes over the columns
template <int WHAT>
__global__ __launch_bounds__(256) void skb_rows_kernel(SkbCtx A, int it) {
    const int pair = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N = A.N, M = A.M, T = A.T, R = N + 1, Cn = M + 1;
    if (i >= R) return;
    const SkbVec v = skb_vec(A, pair);
    const double lmu_i = i < N ? v.norm : v.lmuN;
    const float* __restrict__ G = A.G + (size_t)pair * N * M;
    if (WHAT == SKB_U) {                                         // u^t_i = log mu_i - LSE_j(Z_ij + v^{t-1}_j)
        const double* vp = v.V + (size_t)(it - 1) * Cn;
        double mx = -INFINITY, s = 0;
        for (int j = lane; j < Cn; j += 64) {
            const double x = skb_z(A, pair, i, j, v.a) + vp[j];
            if (x > mx) { s = s * xexp(mx - x) + 1.0; mx = x; } else if (x > -INFINITY) s += xexp(x - mx);
        }
        const double gm = wave_max(mx);
        s = (mx > -INFINITY) ? s * xexp(mx - gm) : 0.0;
        s = wave_sum(s);
        if (lane == 0) v.U[(size_t)(it - 1) * R + i] = lmu_i - (gm + xlog(s));
    } else if (WHAT == SKB_UB) {                                 // ub^t_i = [t == T] sum_j D_ij - sum_j vb^t_j Pv^t_ij
        const double ui = v.U[(size_t)(it - 1) * R + i];
        const double* vv = v.V + (size_t)it * Cn;
        const double* vb = v.VB + (size_t)(it - 1) * Cn;
        double s = 0;
        for (int j = lane; j < Cn; j += 64) {
            const double z = skb_z(A, pair, i, j, v.a);
            if (z > -INFINITY) {
                s -= vb[j] * xexp(z + ui + vv[j] - (j < M ? v.norm : v.lnuM));
                if (it == T && i < N && j < M) s += xexp(z + ui + vv[j] - v.norm) * (double)G[(size_t)i * M + j];
            }
        }
        s = wave_sum(s);
        if (lane == 0) v.UB[(size_t)(it - 1) * R + i] = s;
    } else {                                                     // dL/dZ_ij; the dustbin entries of the row go to GA[i]
        const double* uT = v.U + (size_t)(T - 1) * R;
        const double* vT = v.V + (size_t)T * Cn;
        double ga = 0;
        for (int j = lane; j < Cn; j += 64) {
            const double z = skb_z(A, pair, i, j, v.a);
            double g = 0;
            if (z > -INFINITY) {
                if (i < N && j < M) g = xexp(z + uT[i] + vT[j] - v.norm) * (double)G[(size_t)i * M + j];
                const double lnu_j = j < M ? v.norm : v.lnuM;
                for (int t2 = 1; t2 <= T; ++t2) {
                    const double ui = v.U[(size_t)(t2 - 1) * R + i];
                    g -= v.VB[(size_t)(t2 - 1) * Cn + j] * xexp(z + ui + v.V[(size_t)t2 * Cn + j] - lnu_j);
                    g -= v.UB[(size_t)(t2 - 1) * R + i] * xexp(z + v.V[(size_t)(t2 - 1) * Cn + j] + ui - lmu_i);
                }
            }
            if (i < N && j < M) A.gscores[((size_t)pair * N + i) * M + j] = (float)g;
            else ga += g;
        }
        ga = wave_sum(ga);
        if (lane == 0) v.GA[i] = ga;
    }
}

// column sweeps: 64 columns x SKB_RP = 16 row parts per workgroup (thread (c, p) takes rows p, p + 16, ..: coalesced over c); the partials of a
// column are combined through LDS in the order p = 0 .. 15 (four parts: 40 us per sweep at 375 rows, a chain of 94 dependent steps per thread)
constexpr int SKB_RP = 16;
template <int WHAT>
__global__ __launch_bounds__(64 * SKB_RP) void skb_cols_kernel(SkbCtx A, int it) {
    const int pair = blockIdx.y, c = threadIdx.x & 63, p = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
    const int N = A.N, M = A.M, T = A.T, R = N + 1, Cn = M + 1;
    const SkbVec v = skb_vec(A, pair);
    __shared__ double s_a[SKB_RP][64], s_b[SKB_RP][64];
    const bool live = j < Cn;
    const float* __restrict__ G = A.G + (size_t)pair * N * M;
    double mx = -INFINITY, s = 0;
    if (live) {
        if (WHAT == SKB_V) {                                     // v^t_j = log nu_j - LSE_i(Z_ij + u^t_i)
            const double* un = v.U + (size_t)(it - 1) * R;
            for (int i = p; i < R; i += SKB_RP) {
                const double x = skb_z(A, pair, i, j, v.a) + un[i];
                if (x > mx) { s = s * xexp(mx - x) + 1.0; mx = x; } else if (x > -INFINITY) s += xexp(x - mx);
            }
        } else if (WHAT == SKB_VBT) {                            // vb^T_j = sum_i D_ij
            const double* uT = v.U + (size_t)(T - 1) * R;
            const double vTj = v.V[(size_t)T * Cn + j];
            if (j < M)
                for (int i = p; i < N; i += SKB_RP) {
                    const double z = (double)A.Z[((size_t)pair * N + i) * M + j];
                    if (z > -INFINITY) s += xexp(z + uT[i] + vTj - v.norm) * (double)G[(size_t)i * M + j];
                }
        } else {                                                 // vb^{t-1}_j = - sum_i ub^t_i Pu^t_ij
            const double* u = v.U + (size_t)(it - 1) * R;
            const double* ub = v.UB + (size_t)(it - 1) * R;
            const double vpj = v.V[(size_t)(it - 1) * Cn + j];
            for (int i = p; i < R; i += SKB_RP) {
                const double z = skb_z(A, pair, i, j, v.a);
                if (z > -INFINITY) s -= ub[i] * xexp(z + vpj + u[i] - (i < N ? v.norm : v.lmuN));
            }
        }
    }
    s_a[p][c] = s; s_b[p][c] = mx;
    __syncthreads();
    if (p == 0 && live) {
        if (WHAT == SKB_V) {
            double gm = s_b[0][c];
            for (int k = 1; k < SKB_RP; ++k) gm = s_b[k][c] > gm ? s_b[k][c] : gm;
            double tot = 0;
            for (int k = 0; k < SKB_RP; ++k) tot += s_b[k][c] > -INFINITY ? s_a[k][c] * xexp(s_b[k][c] - gm) : 0.0;
            v.V[(size_t)it * Cn + j] = (j < M ? v.norm : v.lnuM) - (gm + xlog(tot));
        } else {
            double tot = 0;
            for (int k = 0; k < SKB_RP; ++k) tot += s_a[k][c];
            if (WHAT == SKB_VBT) v.VB[(size_t)(T - 1) * Cn + j] = tot;
            else v.VB[(size_t)(it - 2) * Cn + j] = tot;
        }
    }
}

// dL/dalpha of a pair: the rows' dustbin partials in row order
__global__ __launch_bounds__(64) void skb_galpha_kernel(SkbCtx A) {
    const int pair = blockIdx.x;
    if (threadIdx.x != 0) return;
    const SkbVec v = skb_vec(A, pair);
    double s = 0;
    for (int i = 0; i <= A.N; ++i) s += v.GA[i];
    A.galpha[pair] = (float)s;
}

// d loss / d conf of compute_correspondence_loss (sinkhorn form): the clamp passes no gradient outside [1e-6, 1 - 1e-6]
__global__ __launch_bounds__(256) void focal_backward_kernel(long long n, const float* __restrict__ conf, const float* __restrict__ gt,
                                                             const double* __restrict__ part, float alpha, float gamma, float pos_w, float neg_w,
                                                             float* __restrict__ gconf) {
    __shared__ float s_inv[2];
    if (threadIdx.x == 0) {
        double np = 0, nn = 0;
        for (int k = 0; k < TR_BLOCKS; ++k) { np += part[4 * k + 1]; nn += part[4 * k + 3]; }
        s_inv[0] = np > 0 ? pos_w / (float)np : 0.f;
        s_inv[1] = nn > 0 ? neg_w / (float)nn : 0.f;
    }
    __syncthreads();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float c = conf[e], g = gt[e];
    float d = 0.f;
    if (c >= 1e-6f && c <= 1.0f - 1e-6f) {
        if (g == 1.0f) d = -alpha * (-gamma * powg(1.0f - c, gamma - 1.0f) * logf(c) + powg(1.0f - c, gamma) / c) * s_inv[0];
        else if (g == 0.0f) d = -alpha * (gamma * powg(c, gamma - 1.0f) * logf(1.0f - c) - powg(c, gamma) / (1.0f - c)) * s_inv[1];
    }
    gconf[e] = d;
}

}  // namespace
}  // namespace dr

extern "C" {

size_t dr_sinkhorn_backward_workspace_bytes(int P, int N, int M, int iters) {
    if (P < 0 || N < 1 || M < 1 || iters < 1) return 0;
    // per pair: a header of 8, u [T][N+1], v [T+1][M+1], ub [T][N+1], vb [T][M+1], the rows' dustbin partials [N+1] -- doubles
    return (size_t)P * (8 + (size_t)2 * iters * (N + 1) + (size_t)(2 * iters + 1) * (M + 1) + (size_t)(N + 1)) * sizeof(double);
}

int dr_sinkhorn_backward_f32(int P, int N, int M, const float* scores, const uint8_t* src_mask, const uint8_t* tgt_mask, const float* bin_score,
                             int iters, const float* grad_conf, float* grad_scores, float* grad_bin_score, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (P < 0 || N < 1 || M < 1 || iters < 1 || !scores || !bin_score || !grad_conf || !grad_scores || !grad_bin_score || !workspace) return DR_EINVAL;
    if ((src_mask == nullptr) != (tgt_mask == nullptr)) return DR_EINVAL;
    if (workspace_bytes < dr_sinkhorn_backward_workspace_bytes(P, N, M, iters)) return DR_EWORKSPACE;
    if (P == 0) return DR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dr::env_knob("DR_SKB_ONE_WG", 0)) {                  // (diagnostics: the single-workgroup form, its vectors at the head of the workspace)
        dr::SkBwdArgs A1{scores, src_mask, tgt_mask, bin_score, grad_conf, grad_scores, grad_bin_score, workspace, N, M, iters};
        hipLaunchKernelGGL(dr::sk_backward_kernel<double>, dim3(P), dim3(1024), 0, st, A1);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    const int T = iters, R = N + 1, Cn = M + 1;
    dr::SkbCtx A{scores, grad_conf, src_mask, tgt_mask, bin_score, (double*)workspace,
                 8 + (size_t)2 * T * R + (size_t)(2 * T + 1) * Cn + (size_t)R, grad_scores, grad_bin_score, N, M, T};
    const dim3 grows((R + 3) / 4, P), gcols((Cn + 63) / 64, P);
    hipLaunchKernelGGL(dr::skb_prep_kernel, dim3(P), dim3(256), 0, st, A);
    DR_LAUNCH_CHECK();
    for (int it = 1; it <= T; ++it) {                        // forward, keeping every u^t, v^t
        hipLaunchKernelGGL(dr::skb_rows_kernel<dr::SKB_U>, grows, dim3(256), 0, st, A, it);
        hipLaunchKernelGGL(dr::skb_cols_kernel<dr::SKB_V>, gcols, dim3(64 * dr::SKB_RP), 0, st, A, it);
    }
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::skb_cols_kernel<dr::SKB_VBT>, gcols, dim3(64 * dr::SKB_RP), 0, st, A, T);
    for (int it = T; it >= 1; --it) {                        // the adjoint vectors
        hipLaunchKernelGGL(dr::skb_rows_kernel<dr::SKB_UB>, grows, dim3(256), 0, st, A, it);
        if (it > 1) hipLaunchKernelGGL(dr::skb_cols_kernel<dr::SKB_VB>, gcols, dim3(64 * dr::SKB_RP), 0, st, A, it);
    }
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::skb_rows_kernel<dr::SKB_FINAL>, grows, dim3(256), 0, st, A, T);
    hipLaunchKernelGGL(dr::skb_galpha_kernel, dim3(P), dim3(64), 0, st, A);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_focal_loss_backward_f32(int P, int N, int M, const float* conf, const float* conf_gt, float alpha, float gamma, float pos_w, float neg_w,
                               float* grad_conf, void* workspace, void* stream) {
    if (P < 1 || N < 1 || M < 1 || !conf || !conf_gt || !grad_conf || !workspace) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)P * N * M;
    float dummy_unused = 0.f; (void)dummy_unused;
    dr::FocalArgs A{conf, conf_gt, nullptr, n, alpha, gamma, pos_w, neg_w, 0, (double*)workspace, nullptr};
    hipLaunchKernelGGL(dr::focal_partial_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, A);        // class counts
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::focal_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, conf, conf_gt, (const double*)workspace, alpha,
                       gamma, pos_w, neg_w, grad_conf);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // extern "C"

// embed_rotary (3D/models/position_encoding.py:25-35) and its transpose: out = R(+-theta) x * scale, cos / sin [rows, C/2]
namespace dr {
namespace {
__global__ __launch_bounds__(256) void rotary_kernel(long long n2, int halfC, const float* __restrict__ x, const float* __restrict__ cs,
                                                     const float* __restrict__ sn, float sign, float scale, float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;      // pair index: row * C/2 + k
    if (e >= n2) return;
    const float2 v = reinterpret_cast<const float2*>(x)[e];
    const float c = cs[e], s = sign * sn[e];
    float2 o;
    o.x = __fadd_rn(__fmul_rn(v.x, c), __fmul_rn(-v.y, s)) * scale;
    o.y = __fadd_rn(__fmul_rn(v.y, c), __fmul_rn(v.x, s)) * scale;
    reinterpret_cast<float2*>(out)[e] = o;
    (void)halfC;
}
}  // namespace
}  // namespace dr

extern "C" int dr_rotary_f32(int rows, int C, const float* x, const float* cos_t, const float* sin_t, int inverse, float scale, float* out,
                             void* stream) {
    if (rows < 0 || C < 2 || (C & 1) || !x || !cos_t || !sin_t || !out) return DR_EINVAL;
    if (rows == 0) return DR_OK;
    const long long n2 = (long long)rows * (C / 2);
    hipLaunchKernelGGL(dr::rotary_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n2, C / 2, x, cos_t, sin_t,
                       inverse ? -1.0f : 1.0f, scale, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ===============================================================================================================================
// Pieces of the GeometryAttentionLayer backward (3D/models/transformero.py:43-96) that are not GEMMs: LayerNorm forward with saved
// statistics + backward, the masked row softmax of the explicit attention matrix + its backward, ReLU backward.  The GEMMs of the
// backward run on dr_linear_f32; diffreg_hip/autograd.py composes them.  One wave per row; parameter gradients of LayerNorm through a
// fixed grid of partials (bit-reproducible).
// ===============================================================================================================================
namespace dr {
namespace {

__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int C, const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                     float eps, float* __restrict__ y, float* __restrict__ stats) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = (xr[c] - mean) * rstd * g[c] + b[c];
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

constexpr int LNB_BLOCKS = 256;
// gx = rstd (g gy - mean_c(g gy) - xhat mean_c(g gy xhat));  partial sums of gy xhat and gy over this block's rows
__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int C, const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ stats,
                                                     const float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ part) {
    extern __shared__ float sp[];                      // [4 waves][2][C]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* pg = sp + (size_t)w * 2 * C;
    float* pb = pg + C;
    for (int c = lane; c < C; c += 64) { pg[c] = 0.f; pb[c] = 0.f; }
    for (int row = blockIdx.x * 4 + w; row < rows; row += LNB_BLOCKS * 4) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        const float* xr = x + (size_t)row * C;
        const float* gr = gy + (size_t)row * C;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xh = (xr[c] - mean) * rstd, d = gr[c] * g[c];
            s1 += d; s2 = fmaf(d, xh, s2);
            pg[c] = fmaf(gr[c], xh, pg[c]); pb[c] += gr[c];
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
        for (int c = lane; c < C; c += 64) {
            const float xh = (xr[c] - mean) * rstd;
            gx[(size_t)row * C + c] = rstd * (gr[c] * g[c] - s1 - xh * s2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const float v = sp[c] + sp[2 * C + c] + sp[4 * C + c] + sp[6 * C + c];
        part[(size_t)blockIdx.x * 2 * C + c] = v;
    }
}
__global__ __launch_bounds__(256) void ln_bwd_final_kernel(int C, const float* __restrict__ part, float* __restrict__ ggamma, float* __restrict__ gbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * C) return;
    float s = 0.f;
    for (int k = 0; k < LNB_BLOCKS; ++k) s += part[(size_t)k * 2 * C + c];
    if (c < C) ggamma[c] = s; else gbeta[c - C] = s;
}

// P[r][:] = softmax(scale s[r][:]) over the keys; rows r = ((b H + h) L + l); a key j of batch b is masked (-inf) when q_mask[b][l] && !k_mask[b][j]
__global__ __launch_bounds__(256) void softmax_rows_kernel(int rows, int cols, int L, int H, const float* __restrict__ s, float scale,
                                                           const uint8_t* __restrict__ qm, const uint8_t* __restrict__ km, float* __restrict__ P) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int l = row % L, b = row / (L * H);
    const bool qv = qm ? qm[(size_t)b * L + l] != 0 : true;
    const float* sr = s + (size_t)row * cols;
    float mx = -INFINITY;
    for (int j = lane; j < cols; j += 64) {
        const bool dead = km && qv && !km[(size_t)b * cols + j];
        if (!dead) mx = fmaxf(mx, sr[j] * scale);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const bool dead = km && qv && !km[(size_t)b * cols + j];
        if (!dead) sum += expf(sr[j] * scale - mx);
    }
    sum = wave_sum(sum);
    for (int j = lane; j < cols; j += 64) {
        const bool dead = km && qv && !km[(size_t)b * cols + j];
        P[(size_t)row * cols + j] = dead ? 0.f : expf(sr[j] * scale - mx) / sum;
    }
}
// dual-softmax read-out of Matching.forward (3D/models/matching.py:193-205): conf = softmax over the ROWS (dim 1) of sim / T with the invalid source
// rows at -inf, times softmax over the COLUMNS (dim 2) with the invalid target columns at -inf.  Column statistics first (one thread per column, rows in
// order), then one wave per row.
__global__ __launch_bounds__(256) void dual_softmax_cols_kernel(int N, int M, const float* __restrict__ sim, float T, const uint8_t* __restrict__ sm,
                                                                float* __restrict__ cmax, float* __restrict__ csum) {
    const int p = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const float* s = sim + (size_t)p * N * M + j;
    const uint8_t* m = sm ? sm + (size_t)p * N : nullptr;
    float mx = -INFINITY;
    for (int i = 0; i < N; ++i)
        if (!m || m[i]) mx = fmaxf(mx, s[(size_t)i * M] / T);
    float sum = 0.f;
    for (int i = 0; i < N; ++i)
        if (!m || m[i]) sum += expf(s[(size_t)i * M] / T - mx);
    cmax[(size_t)p * M + j] = mx;
    csum[(size_t)p * M + j] = sum;
}
__global__ __launch_bounds__(256) void dual_softmax_rows_kernel(int rows, int N, int M, const float* __restrict__ sim, float T, const uint8_t* __restrict__ sm,
                                                                const uint8_t* __restrict__ tm, const float* __restrict__ cmax,
                                                                const float* __restrict__ csum, float* __restrict__ conf) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int p = row / N;
    const bool rv = sm ? sm[row] != 0 : true;
    const float* sr = sim + (size_t)row * M;
    const uint8_t* t = tm ? tm + (size_t)p * M : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < M; j += 64)
        if (!t || t[j]) mx = fmaxf(mx, sr[j] / T);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < M; j += 64)
        if (!t || t[j]) sum += expf(sr[j] / T - mx);
    sum = wave_sum(sum);
    for (int j = lane; j < M; j += 64) {
        const float x = sr[j] / T;
        const float a = rv ? expf(x - cmax[(size_t)p * M + j]) / csum[(size_t)p * M + j] : 0.f;
        const float b = (!t || t[j]) ? expf(x - mx) / sum : 0.f;
        conf[(size_t)row * M + j] = a * b;
    }
}
// ---- backward of the dual-softmax read-out.  x = sim / T, A = softmax over the valid rows of a column, B = softmax over the valid columns of a row,
// conf = A B, u = grad_conf conf:  d loss / d x = 2 u - A colsum(u) - B rowsum(u)  (the two softmax adjoints; masked entries have A = 0 or B = 0 and
// receive exactly what torch's masked_fill_ lets through: nothing).  Row pass (statistics + rowsum), column pass (colsum, rows in order), final row pass.
__global__ __launch_bounds__(256) void dsm_bwd_rows_kernel(int rows, int N, int M, const float* __restrict__ sim, float T, const uint8_t* __restrict__ sm,
                                                           const uint8_t* __restrict__ tm, const float* __restrict__ cmax, const float* __restrict__ csum,
                                                           const float* __restrict__ g, float* __restrict__ rmax, float* __restrict__ rsum,
                                                           float* __restrict__ rs) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int p = row / N;
    const bool rv = sm ? sm[row] != 0 : true;
    const float* sr = sim + (size_t)row * M;
    const uint8_t* t = tm ? tm + (size_t)p * M : nullptr;
    float mx = -INFINITY;
    for (int j = lane; j < M; j += 64)
        if (!t || t[j]) mx = fmaxf(mx, sr[j] / T);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < M; j += 64)
        if (!t || t[j]) sum += expf(sr[j] / T - mx);
    sum = wave_sum(sum);
    float u = 0.f;
    for (int j = lane; j < M; j += 64) {
        const float x = sr[j] / T;
        const float a = rv ? expf(x - cmax[(size_t)p * M + j]) / csum[(size_t)p * M + j] : 0.f;
        const float b = (!t || t[j]) ? expf(x - mx) / sum : 0.f;
        u += g[(size_t)row * M + j] * (a * b);
    }
    u = wave_sum(u);
    if (lane == 0) { rmax[row] = mx; rsum[row] = sum; rs[row] = u; }
}
__global__ __launch_bounds__(256) void dsm_bwd_cols_kernel(int N, int M, const float* __restrict__ sim, float T, const uint8_t* __restrict__ sm,
                                                           const uint8_t* __restrict__ tm, const float* __restrict__ cmax, const float* __restrict__ csum,
                                                           const float* __restrict__ rmax, const float* __restrict__ rsum, const float* __restrict__ g,
                                                           float* __restrict__ cs) {
    const int p = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const float* s = sim + (size_t)p * N * M + j;
    const float* gg = g + (size_t)p * N * M + j;
    const uint8_t* m = sm ? sm + (size_t)p * N : nullptr;
    const bool cv = tm ? tm[(size_t)p * M + j] != 0 : true;
    const float cm = cmax[(size_t)p * M + j], cq = csum[(size_t)p * M + j];
    float acc = 0.f;
    if (cv)
        for (int i = 0; i < N; ++i) {
            if (m && !m[i]) continue;
            const float x = s[(size_t)i * M] / T;
            const float a = expf(x - cm) / cq, b = expf(x - rmax[(size_t)p * N + i]) / rsum[(size_t)p * N + i];
            acc += gg[(size_t)i * M] * (a * b);
        }
    cs[(size_t)p * M + j] = acc;
}
__global__ __launch_bounds__(256) void dsm_bwd_final_kernel(int rows, int N, int M, const float* __restrict__ sim, float T, const uint8_t* __restrict__ sm,
                                                            const uint8_t* __restrict__ tm, const float* __restrict__ cmax, const float* __restrict__ csum,
                                                            const float* __restrict__ rmax, const float* __restrict__ rsum, const float* __restrict__ rs,
                                                            const float* __restrict__ cs, const float* __restrict__ g, float* __restrict__ gsim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int p = row / N;
    const bool rv = sm ? sm[row] != 0 : true;
    const float* sr = sim + (size_t)row * M;
    const uint8_t* t = tm ? tm + (size_t)p * M : nullptr;
    const float mx = rmax[row], sum = rsum[row], r = rs[row];
    for (int j = lane; j < M; j += 64) {
        const float x = sr[j] / T;
        const float a = rv ? expf(x - cmax[(size_t)p * M + j]) / csum[(size_t)p * M + j] : 0.f;
        const float b = (!t || t[j]) ? expf(x - mx) / sum : 0.f;
        const float u = g[(size_t)row * M + j] * (a * b);
        gsim[(size_t)row * M + j] = (2.f * u - a * cs[(size_t)p * M + j] - b * r) / T;
    }
}
// dS = scale P (dP - sum_j dP_j P_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int rows, int cols, const float* __restrict__ P, const float* __restrict__ dP, float scale,
                                                          float* __restrict__ dS) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = P + (size_t)row * cols;
    const float* d = dP + (size_t)row * cols;
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) s = fmaf(p[j], d[j], s);
    s = wave_sum(s);
    for (int j = lane; j < cols; j += 64) dS[(size_t)row * cols + j] = scale * p[j] * (d[j] - s);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(long long n, const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < n) gx[e] = y[e] > 0.f ? gy[e] : 0.f;
}

}  // namespace
}  // namespace dr

extern "C" {

int dr_layernorm_f32(int rows, int C, const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean_rstd, void* stream) {
    if (rows < 0 || C < 1 || !x || !gamma || !beta || !y || !mean_rstd) return DR_EINVAL;
    if (rows == 0) return DR_OK;
    hipLaunchKernelGGL(dr::ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, C, x, gamma, beta, eps, y, mean_rstd);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
size_t dr_layernorm_backward_workspace_bytes(int C) { return C > 0 ? (size_t)dr::LNB_BLOCKS * 2 * C * sizeof(float) : 0; }
int dr_layernorm_backward_f32(int rows, int C, const float* x, const float* gamma, const float* mean_rstd, const float* grad_y, float* grad_x,
                              float* grad_gamma, float* grad_beta, void* workspace, void* stream) {
    if (rows < 1 || C < 1 || C > 2048 || !x || !gamma || !mean_rstd || !grad_y || !grad_x || !grad_gamma || !grad_beta || !workspace) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dr::ln_bwd_kernel, dim3(dr::LNB_BLOCKS), dim3(256), (size_t)8 * C * sizeof(float), st, rows, C, x, gamma, mean_rstd, grad_y, grad_x,
                       (float*)workspace);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::ln_bwd_final_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, C, (const float*)workspace, grad_gamma, grad_beta);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int dr_softmax_rows_f32(int B, int H, int L, int S, const float* scores, float scale, const uint8_t* q_mask, const uint8_t* k_mask, float* P, void* stream) {
    if (B < 0 || H < 1 || L < 1 || S < 1 || !scores || !P || ((q_mask == nullptr) != (k_mask == nullptr))) return DR_EINVAL;
    const int rows = B * H * L;
    if (rows == 0) return DR_OK;
    hipLaunchKernelGGL(dr::softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, S, L, H, scores, scale, q_mask, k_mask, P);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int dr_dual_softmax_f32(int P, int N, int M, const float* sim, float temperature, const uint8_t* src_mask, const uint8_t* tgt_mask, float* conf,
                        float* col_stats, void* stream) {
    if (P < 0 || N < 1 || M < 1 || !sim || !conf || !col_stats || !(temperature > 0.f) || ((src_mask == nullptr) != (tgt_mask == nullptr))) return DR_EINVAL;
    if (P == 0) return DR_OK;
    float* cmax = col_stats;
    float* csum = col_stats + (size_t)P * M;
    hipLaunchKernelGGL(dr::dual_softmax_cols_kernel, dim3((M + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, N, M, sim, temperature, src_mask, cmax, csum);
    DR_LAUNCH_CHECK();
    const int rows = P * N;
    hipLaunchKernelGGL(dr::dual_softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, N, M, sim, temperature, src_mask, tgt_mask,
                       cmax, csum, conf);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
size_t dr_dual_softmax_backward_workspace_bytes(int P, int N, int M) { return (P > 0 && N > 0 && M > 0) ? (size_t)P * 3 * ((size_t)N + M) * sizeof(float) : 0; }
int dr_dual_softmax_backward_f32(int P, int N, int M, const float* sim, float temperature, const uint8_t* src_mask, const uint8_t* tgt_mask,
                                 const float* grad_conf, float* grad_sim, void* workspace, size_t workspace_bytes, void* stream) {
    if (P < 0 || N < 1 || M < 1 || !sim || !grad_conf || !grad_sim || !(temperature > 0.f) || ((src_mask == nullptr) != (tgt_mask == nullptr))) return DR_EINVAL;
    if (P == 0) return DR_OK;
    if (!workspace || workspace_bytes < dr_dual_softmax_backward_workspace_bytes(P, N, M)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* cmax = (float*)workspace;
    float *csum = cmax + (size_t)P * M, *cs = csum + (size_t)P * M, *rmax = cs + (size_t)P * M, *rsum = rmax + (size_t)P * N, *rs = rsum + (size_t)P * N;
    const int rows = P * N;
    hipLaunchKernelGGL(dr::dual_softmax_cols_kernel, dim3((M + 255) / 256, P), dim3(256), 0, st, N, M, sim, temperature, src_mask, cmax, csum);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::dsm_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, N, M, sim, temperature, src_mask, tgt_mask, cmax, csum, grad_conf, rmax,
                       rsum, rs);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::dsm_bwd_cols_kernel, dim3((M + 255) / 256, P), dim3(256), 0, st, N, M, sim, temperature, src_mask, tgt_mask, cmax, csum, rmax, rsum,
                       grad_conf, cs);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::dsm_bwd_final_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, N, M, sim, temperature, src_mask, tgt_mask, cmax, csum, rmax, rsum, rs,
                       cs, grad_conf, grad_sim);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int dr_softmax_backward_f32(int rows, int cols, const float* P, const float* grad_P, float scale, float* grad_scores, void* stream) {
    if (rows < 0 || cols < 1 || !P || !grad_P || !grad_scores) return DR_EINVAL;
    if (rows == 0) return DR_OK;
    hipLaunchKernelGGL(dr::softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, cols, P, grad_P, scale, grad_scores);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int dr_relu_backward_f32(long long n, const float* y, const float* grad_y, float* grad_x, void* stream) {
    if (n < 0 || !y || !grad_y || !grad_x) return DR_EINVAL;
    if (n == 0) return DR_OK;
    hipLaunchKernelGGL(dr::relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, y, grad_y, grad_x);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // extern "C"

// d loss / d (R_pred, t_pred) of the L1 motion term (loss.py:108-128): e1 = sum_c |(R_p s + t_p)_c - (R_g s' + t_g)_c| averaged over the overlap rows of
// the whole batch: dL/dR_p[b][c][j] = sum_i sign(d_ic) s_ij / n, dL/dt_p[b][c] = sum_i sign(d_ic) / n.  One workgroup per pair, rows in a fixed order.
namespace dr {
namespace {
__global__ __launch_bounds__(256) void motion_bwd_kernel(MotionArgs A, const double* __restrict__ part, float* __restrict__ gR, float* __restrict__ gt) {
    __shared__ float s_acc[4][12];
    __shared__ float s_n;
    const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        double cnt = 0;
        for (int k = 0; k < TR_BLOCKS; ++k) cnt += part[2 * k + 1];
        s_n = (float)cnt;
    }
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    const float* Rp = A.Rp + 9 * b; const float* Rg = A.Rg + 9 * b;
    for (int i = threadIdx.x; i < A.N; i += 256) {
        const size_t e = (size_t)b * A.N + i;
        if (!A.mask[e]) continue;
        const float x = A.s[3 * e], y = A.s[3 * e + 1], z = A.s[3 * e + 2];
        float dx = x, dy = y, dz = z;
        if (A.flow) { dx += A.flow[3 * e]; dy += A.flow[3 * e + 1]; dz += A.flow[3 * e + 2]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float wp = (Rp[3 * c] * x + Rp[3 * c + 1] * y + Rp[3 * c + 2] * z) + A.tp[3 * b + c];
            const float wg = (Rg[3 * c] * dx + Rg[3 * c + 1] * dy + Rg[3 * c + 2] * dz) + A.tg[3 * b + c];
            const float sc = c == 0 ? x : (c == 1 ? y : z);
            const float d = (wp - sc) - (wg - sc);
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            acc[3 * c] += sg * x; acc[3 * c + 1] += sg * y; acc[3 * c + 2] += sg * z; acc[9 + c] += sg;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) { const float v = wave_sum(acc[k]); if (lane == 0) s_acc[w][k] = v; }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float v = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
        if (threadIdx.x < 9) gR[9 * b + threadIdx.x] = v / s_n; else gt[3 * b + threadIdx.x - 9] = v / s_n;
    }
}
}  // namespace
}  // namespace dr

extern "C" int dr_motion_l1_backward_f32(int P, int N, const float* s_pcd, const float* flow, const float* R_pred, const float* t_pred, const float* R_gt,
                                         const float* t_gt, const uint8_t* overlap_mask, float* grad_R, float* grad_t, void* workspace, void* stream) {
    if (P < 1 || N < 1 || !s_pcd || !R_pred || !t_pred || !R_gt || !t_gt || !overlap_mask || !grad_R || !grad_t || !workspace) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dr::MotionArgs A{s_pcd, flow, R_pred, t_pred, R_gt, t_gt, overlap_mask, P, N, (double*)workspace, nullptr};
    hipLaunchKernelGGL(dr::motion_partial_kernel, dim3(dr::TR_BLOCKS), dim3(dr::TR_THREADS), 0, st, A);       // the number of overlap rows
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dr::motion_bwd_kernel, dim3(P), dim3(256), 0, st, A, (const double*)workspace, grad_R, grad_t);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
