// fine2d3d.hip -- the patch-correspondence block behind the 2D-3D loop (EXP = Diff-Reg-2d3d/experiments/
// 2d3dmatr.rgbdv2.stage4.level3.stage1): EXP/model.py:707-780.
//   patch_similarity   index_select of the two patches' fine features + pairwise_cosine_similarity(normalized = True)
//                      (model.py:726-738; vision3d/ops/index_select.py:4-33, cosine_similarity.py:34-66)
//   unique_i64         duplicate removal: torch.unique of img_index * num_points_f + pcd_index (model.py:760-763)
//   corr_gather        the final gathers and corr_scores = <img_feats_f[i], pcd_feats_f[j]> (model.py:766-774)
// (the selection between them is dr_mutual_topk_select_f32, stateops.hip).  Off the hot path: small, memory- / latency-bound
// kernels; unique_i64 = the bitonic (key, index) sort of collate.hip + a one-workgroup ordered compaction of the segment heads.
#include <cstring>
#include "kernels.h"

namespace dr {
namespace {

// one workgroup = one correspondence b x one 64-row tile of its image patch: S[i][j] = 0.5 (<x_i, y_j> + 1) for the Kc patch points.
// Feature chunks of 32 go through LDS ([rows][33]); a thread owns rows 4 ty .. + 3 and columns tx + 32 jj (bank-conflict-free reads).
constexpr int PS_TI = 64, PS_KC = 32, PS_MAXKC = 128;

__global__ __launch_bounds__(256) void patch_similarity_kernel(int Ki, int Kc, int C, const float* __restrict__ img_feats, const long long* __restrict__ img_idx,
                                                               const float* __restrict__ pcd_feats, const long long* __restrict__ pcd_idx,
                                                               long long pcd_rows, float* __restrict__ out) {
    __shared__ float sx[PS_TI][PS_KC + 1];
    __shared__ float sy[PS_MAXKC][PS_KC + 1];
    __shared__ long long si[PS_TI], sj[PS_MAXKC];
    const int b = blockIdx.y, i0 = blockIdx.x * PS_TI, t = threadIdx.x, tx = t & 31, ty = t >> 5;   // ty 0..7
    if (t < PS_TI) si[t] = i0 + t < Ki ? img_idx[(size_t)b * Ki + i0 + t] : -1;
    if (t < PS_MAXKC) sj[t] = t < Kc ? pcd_idx[(size_t)b * Kc + t] : -1;
    __syncthreads();
    float acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
    for (int k0 = 0; k0 < C; k0 += PS_KC) {
        for (int e = t; e < PS_TI * PS_KC; e += 256) {
            const int r = e / PS_KC, k = e % PS_KC;
            sx[r][k] = (si[r] >= 0 && k0 + k < C) ? img_feats[(size_t)si[r] * C + k0 + k] : 0.f;
        }
        for (int e = t; e < PS_MAXKC * PS_KC; e += 256) {
            const int r = e / PS_KC, k = e % PS_KC;
            // (row `pcd_rows` of the padded feature matrix is the zero row of model.py:707)
            sy[r][k] = (sj[r] >= 0 && sj[r] < pcd_rows && k0 + k < C) ? pcd_feats[(size_t)sj[r] * C + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < PS_KC; ++k) {
            float xv[8], yv[4];
#pragma unroll
            for (int r = 0; r < 8; ++r) xv[r] = sx[8 * ty + r][k];
#pragma unroll
            for (int c = 0; c < 4; ++c) yv[c] = sy[tx + 32 * c][k];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(xv[r], yv[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = i0 + 8 * ty + r;
        if (i >= Ki) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = tx + 32 * c;
            if (j < Kc) out[((size_t)b * Ki + i) * Kc + j] = 0.5f * (acc[r][c] + 1.0f);
        }
    }
}

// keys as unsigned 64-bit numbers in the order of the signed ones (sign bit flipped); rows >= n: pad keys that sort last
constexpr unsigned long long UQ_BIAS = 0x8000000000000000ull;
__global__ __launch_bounds__(256) void make_keys_kernel(int n, int n_pad, const long long* __restrict__ a, const long long* __restrict__ b,
                                                        long long mul, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    keys[i] = i < n ? ((unsigned long long)(a[i] * mul + b[i]) ^ UQ_BIAS) : ~0ull;
    vals[i] = (unsigned)i;
}
// the distinct keys of the sorted list in ascending order (torch.unique): ONE workgroup walks the list 1024 entries at a time; an
// entry is kept when it differs from its predecessor; ranks inside a step = wave ballots + a 16-entry scan (lists of this block
// are a few thousand entries: latency of one small kernel, no multi-kernel scan)
__global__ __launch_bounds__(1024) void unique_sorted_kernel(int n, const unsigned long long* __restrict__ sorted, long long* __restrict__ out,
                                                             int* __restrict__ count) {
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + t;
        const unsigned long long k = i < n ? sorted[i] : 0ull;
        const bool head = i < n && (i == 0 || sorted[i - 1] != k);
        const unsigned long long b = __ballot(head);
        if (lane == 0) s_w[w] = __popcll(b);
        __syncthreads();
        int before = s_base;
        for (int q = 0; q < w; ++q) before += s_w[q];
        if (head) out[before + __popcll(b & ((1ull << lane) - 1ull))] = (long long)(k ^ UQ_BIAS);
        __syncthreads();
        if (t == 0) { int tot = 0; for (int q = 0; q < 16; ++q) tot += s_w[q]; s_base += tot; }
        __syncthreads();
    }
    if (t == 0) *count = s_base;
}

// one wave per correspondence: indices back from the key, the point / pixel rows, the feature dot product
__global__ __launch_bounds__(256) void corr_gather_kernel(const int* __restrict__ count, const long long* __restrict__ keys, long long num_points_f, int C,
                                                          const float* __restrict__ img_points, const float* __restrict__ img_pixels,
                                                          const float* __restrict__ pcd_points, const float* __restrict__ pcd_pixels,
                                                          const float* __restrict__ img_feats, const float* __restrict__ pcd_feats,
                                                          long long* __restrict__ img_corr_idx, long long* __restrict__ pcd_corr_idx,
                                                          float* __restrict__ o_img_pts, float* __restrict__ o_img_pix, float* __restrict__ o_pcd_pts,
                                                          float* __restrict__ o_pcd_pix, float* __restrict__ scores) {
    const int n = *count;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= n) return;
    const long long key = keys[c], ii = key / num_points_f, pj = key % num_points_f;
    float s = 0.f;
    for (int k = lane; k < C; k += 64) s = fmaf(img_feats[(size_t)ii * C + k], pcd_feats[(size_t)pj * C + k], s);
    s = wave_sum(s);
    if (lane == 0) { img_corr_idx[c] = ii; pcd_corr_idx[c] = pj; scores[c] = s; }
    if (lane < 3) { o_img_pts[3 * c + lane] = img_points[3 * ii + lane]; o_pcd_pts[3 * c + lane] = pcd_points[3 * pj + lane]; }
    if (lane < 2) { o_img_pix[2 * c + lane] = img_pixels[2 * ii + lane]; o_pcd_pix[2 * c + lane] = pcd_pixels[2 * pj + lane]; }
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace
}  // namespace dr

extern "C" {

int dr_patch_similarity_f32(int P, int Ki, int Kc, int C, const float* img_feats, const int64_t* img_knn_indices, const float* pcd_feats,
                            const int64_t* pcd_knn_indices, long long pcd_rows, float* out, void* stream) {
    if (P < 0 || Ki < 1 || Kc < 1 || Kc > dr::PS_MAXKC || C < 1 || !img_feats || !img_knn_indices || !pcd_feats || !pcd_knn_indices || !out)
        return Kc > dr::PS_MAXKC ? DR_ENOSUP : DR_EINVAL;
    if (P == 0) return DR_OK;
    hipLaunchKernelGGL(dr::patch_similarity_kernel, dim3((Ki + dr::PS_TI - 1) / dr::PS_TI, P), dim3(256), 0, (hipStream_t)stream, Ki, Kc, C, img_feats,
                       (const long long*)img_knn_indices, pcd_feats, (const long long*)pcd_knn_indices, pcd_rows, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

static int uq_pad(int n) { int p = 1; while (p < n) p <<= 1; return p; }

size_t dr_unique_pairs_workspace_bytes(int n) {
    if (n < 1) return 256;
    const size_t np = (size_t)uq_pad(n);
    return dr::align256(np * 8) + dr::align256(np * 4);
}

int dr_unique_pairs_i64(int n, const int64_t* first, const int64_t* second, long long multiplier, int64_t* unique_keys, int32_t* count,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || !unique_keys || !count || !workspace || (n > 0 && (!first || !second))) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { DR_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int32_t), st)); return DR_OK; }
    if (workspace_bytes < dr_unique_pairs_workspace_bytes(n)) return DR_EWORKSPACE;
    const int n_pad = uq_pad(n);
    unsigned long long* keys = (unsigned long long*)workspace;
    unsigned* vals = (unsigned*)((char*)workspace + dr::align256((size_t)n_pad * 8));
    hipLaunchKernelGGL(dr::make_keys_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, n, n_pad, (const long long*)first, (const long long*)second,
                       multiplier, keys, vals);
    DR_LAUNCH_CHECK();
    const int rc = dr::launch_bitonic_sort(keys, vals, n_pad, st);
    if (rc) return rc;
    hipLaunchKernelGGL(dr::unique_sorted_kernel, dim3(1), dim3(1024), 0, st, n, keys, (long long*)unique_keys, count);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_corr_gather_f32(int capacity, const int32_t* count, const int64_t* unique_keys, long long num_points_f, int C, const float* img_points_f,
                       const float* img_pixels_f, const float* pcd_points_f, const float* pcd_pixels_f, const float* img_feats_f, const float* pcd_feats_f,
                       int64_t* img_corr_indices, int64_t* pcd_corr_indices, float* img_corr_points, float* img_corr_pixels, float* pcd_corr_points,
                       float* pcd_corr_pixels, float* corr_scores, void* stream) {
    if (capacity < 0 || !count || !unique_keys || num_points_f < 1 || C < 1 || !img_points_f || !img_pixels_f || !pcd_points_f || !pcd_pixels_f ||
        !img_feats_f || !pcd_feats_f || !img_corr_indices || !pcd_corr_indices || !img_corr_points || !img_corr_pixels || !pcd_corr_points ||
        !pcd_corr_pixels || !corr_scores)
        return DR_EINVAL;
    if (capacity == 0) return DR_OK;
    hipLaunchKernelGGL(dr::corr_gather_kernel, dim3((capacity + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const int*)count, (const long long*)unique_keys,
                       num_points_f, C, img_points_f, img_pixels_f, pcd_points_f, pcd_pixels_f, img_feats_f, pcd_feats_f, (long long*)img_corr_indices,
                       (long long*)pcd_corr_indices, img_corr_points, img_corr_pixels, pcd_corr_points, pcd_corr_pixels, corr_scores);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // extern "C"
