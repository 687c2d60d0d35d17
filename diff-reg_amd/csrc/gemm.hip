// gemm.hip -- grouped fp32 "NT" GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32).
//
// out[r][c] = epi(sum_k A[r][k] W[c][k]) : the bias-free nn.Linear layers of the reference
// (q/k/v/merge/mlp of 3D/models/transformero.py:26-37, src_proj of 3D/models/matching.py:107).
// Both operands are K-contiguous, so a tile of each is staged in LDS as [rows][32 k] (+4 pad ->
// conflict-free ds_read_b128) and every lane fetches 4 consecutive k of "its" row per read:
// lane half h = lane >> 5 takes k = 8g + 4h .. 8g + 4h + 3, and MFMA step e multiplies element e of
// both fragments -- the k order inside a group is permuted identically for A and W, which a sum
// over k does not care about.  One ds_read_b128 per operand feeds 4 MFMAs.
// 4 waves per workgroup arranged WM x WN x WK (WK = split of the k range inside the workgroup,
// reduced through LDS) so that small problems (256..512 rows) still spread over the chip.
#include "kernels.h"

namespace dr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;        // k per staged chunk
constexpr int LDT = BK + 4;   // padded LDS row stride (floats)

template <int WM, int WN, int WK>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmBatch G) {
    static_assert(WM * WN * WK == 4, "4 waves");
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int A_SLOTS = BM * 8 / 256 > 0 ? BM * 8 / 256 : 1;   // float4 slots per thread
    constexpr int B_SLOTS = BN * 8 / 256 > 0 ? BN * 8 / 256 : 1;
    constexpr int STAGE = (BM + BN) * LDT;                          // floats per buffer
    constexpr int RED = (WK > 1) ? 4 * 16 * 64 : 0;
    constexpr int SMEM = 2 * STAGE > RED ? 2 * STAGE : RED;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const GemmProblem& P = G.p[blockIdx.y];
    const int tiles_n = (P.ncols + BN - 1) / BN, tiles_m = (P.rows + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wk = w % WK, wn = (w / WK) % WN, wm = w / (WK * WN);
    const int K = P.K, K1 = P.A2 ? P.K1 : P.K;
    const int nchunks = (K + BK - 1) / BK;

    float4 ra[A_SLOTS], rb[B_SLOTS];
    auto load_chunk = [&](int ch) {
        const int k0 = ch * BK;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const int slot = t + s * 256;
            const int r = slot >> 3, k = k0 + 4 * (slot & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < BM * 8 && row0 + r < P.rows && k < K) {
                if (k < K1) v = *reinterpret_cast<const float4*>(P.A + (size_t)(row0 + r) * P.lda + k);
                else v = *reinterpret_cast<const float4*>(P.A2 + (size_t)(row0 + r) * P.lda2 + (k - K1));
            }
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const int slot = t + s * 256;
            const int r = slot >> 3, k = k0 + 4 * (slot & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < BN * 8 && col0 + r < P.ncols && k < K)
                v = *reinterpret_cast<const float4*>(P.W + (size_t)(col0 + r) * K + k);
            rb[s] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + BM * LDT;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            const int slot = t + s * 256;
            if (slot < BM * 8) *reinterpret_cast<float4*>(As + (slot >> 3) * LDT + 4 * (slot & 7)) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const int slot = t + s * 256;
            if (slot < BN * 8) *reinterpret_cast<float4*>(Bs + (slot >> 3) * LDT + 4 * (slot & 7)) = rb[s];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    constexpr int GROUPS = BK / 8 / WK;          // 8-wide k groups per wave per chunk
    const int h = lane >> 5, l31 = lane & 31;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) load_chunk(ch + 1);
        const float* As = smem + (ch & 1) * STAGE + (wm * 32 + l31) * LDT + wk * GROUPS * 8 + 4 * h;
        const float* Bs = smem + (ch & 1) * STAGE + BM * LDT + (wn * 32 + l31) * LDT + wk * GROUPS * 8 + 4 * h;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(As + 8 * g);
            const float4 b = *reinterpret_cast<const float4*>(Bs + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
        __syncthreads();
    }

    if (WK > 1) {
        // reduce the WK partial accumulators of each (wm, wn) through LDS
        float* red = smem;
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(w * 16 + i) * 64 + lane] = acc[i];
        __syncthreads();
        if (wk != 0) return;
#pragma unroll
        for (int o = 1; o < WK; ++o)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += red[((w + o) * 16 + i) * 64 + lane];
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int col = col0 + wn * 32 + l31;
    const bool col_ok = col < P.ncols;
    const int halfC = P.rot_C >> 1;
    const int ridx = (P.epi & EPI_ROTARY) ? (col % P.rot_C) >> 1 : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        float v = acc[i];
        if (P.epi & EPI_ROTARY) {
            // x*cos + swap(x)*sin, swap(x)[2k] = -x[2k+1], swap(x)[2k+1] = x[2k]  (position_encoding.py:25-35)
            const float other = __shfl_xor(v, 1);
            if (row < P.rows && col_ok) {
                const float c = P.cosT[(size_t)row * halfC + ridx], s = P.sinT[(size_t)row * halfC + ridx];
                const float sw = (col & 1) ? other : -other;
                v = __fadd_rn(__fmul_rn(v, c), __fmul_rn(sw, s));
            }
        }
        if (P.epi & EPI_RELU) v = fmaxf(v, 0.f);
        v *= P.scale;
        if (row < P.rows && col_ok) P.out[(size_t)row * P.ldo + col] = v;
    }
}

template <int WM, int WN, int WK>
static int launch_cfg(const GemmBatch& g, hipStream_t st) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    int maxt = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tl = ((g.p[i].rows + BM - 1) / BM) * ((g.p[i].ncols + BN - 1) / BN);
        maxt = tl > maxt ? tl : maxt;
    }
    if (maxt == 0) return DR_OK;
    double flops = 0;
    for (int i = 0; i < g.n; ++i) flops += 2.0 * g.p[i].rows * g.p[i].ncols * g.p[i].K;
    ProfScope ps(PK_GEMM, flops, st);
    hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, WK>), dim3(maxt, g.n), dim3(256), 0, st, g);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_gemm(const GemmBatch& g, hipStream_t st) {
    if (g.n < 1 || g.n > 4) return DR_EINVAL;
    long tiles64 = 0;
    for (int i = 0; i < g.n; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.K % 4 || p.lda % 4 || (p.A2 && (p.K1 % 4 || p.lda2 % 4))) return DR_ENOSUP;
        if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.A2) & 15) return DR_ENOSUP;
        tiles64 += (long)((p.rows + 63) / 64) * ((p.ncols + 63) / 64);
    }
    // fill the 256 CUs: wide tiles when there is enough work, otherwise split k inside the workgroup
    if (tiles64 >= 512) return launch_cfg<2, 2, 1>(g, st);
    if (tiles64 >= 128) return launch_cfg<2, 1, 2>(g, st);
    return launch_cfg<1, 1, 4>(g, st);
}

}  // namespace dr
