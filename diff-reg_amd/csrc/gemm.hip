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
#include <cstdlib>
#include <type_traits>
#include "kernels.h"

namespace dr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Epilogue shared by both kernels.  C/D layout of the 32x32 MFMA (any input type): col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
template <int TM, int TN, class Acc>
__device__ __forceinline__ void gemm_epilogue(const GemmProblem& P, Acc acc, int rows, int ncols, int row0, int col0,
                                              int wm, int wn, int lane, size_t out_off = 0) {
    const int h = lane >> 5, l31 = lane & 31;
    float* __restrict__ outp = P.out + out_off;
    const int halfC = P.rot_C >> 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + (wn * TN + j) * 32 + l31;
        const bool col_ok = col < ncols;
        const int ridx = (P.epi & EPI_ROTARY) ? (col % P.rot_C) >> 1 : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc(i, j)[r];
                if (P.epi & EPI_ROTARY) {
                    // x*cos + swap(x)*sin, swap(x)[2k] = -x[2k+1], swap(x)[2k+1] = x[2k]  (position_encoding.py:25-35)
                    const float other = __shfl_xor(v, 1);
                    if (row < rows && col_ok) {
                        const float c = P.cosT[(size_t)row * halfC + ridx], s = P.sinT[(size_t)row * halfC + ridx];
                        const float sw = (col & 1) ? other : -other;
                        v = __fadd_rn(__fmul_rn(v, c), __fmul_rn(sw, s));
                    }
                }
                if (P.bias && col_ok) v += P.bias[col];
                if (P.epi & EPI_RELU) v = fmaxf(v, 0.f);
                v *= P.scale;
                if (row < rows && col_ok) {
                    if (P.addend) v += P.addend[(size_t)row * P.ldo + col];
                    outp[(size_t)row * P.ldo + col] = v;
                }
            }
    }
}

// Geometry: a wave owns TM x TN MFMA tiles (32x32 each); a workgroup is WM x WN x WK waves (4 in all);
// K is staged in chunks of BKC floats (double-buffered LDS, one barrier per chunk), of which each of
// the WK k-groups of waves consumes BKC / WK.
template <int TM, int TN, int WM, int WN, int WK, int BKC, int NBUF = 2>
struct GemmGeom {
    static constexpr int NW = WM * WN * WK, NT = 64 * NW;       // waves / threads per workgroup
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static constexpr int LDT = BKC + 4;                         // padded LDS row stride (floats)
    static constexpr int C4 = BKC / 4;                          // float4 per staged row
    static constexpr int A_SLOTS = (BM * C4 + NT - 1) / NT, B_SLOTS = (BN * C4 + NT - 1) / NT;
    static constexpr int STAGE = (BM + BN) * LDT;               // floats per buffer
    static constexpr int RED = (WK > 1) ? NW * TM * TN * 16 * 64 : 0;
    static constexpr int SMEM_FLOATS = NBUF * STAGE > RED ? NBUF * STAGE : RED;
    static constexpr int GROUPS = BKC / WK / 8;                 // 8-wide k groups per wave per chunk
    static constexpr int NACC = (TM * TN == 1) ? 2 : 1;         // independent accumulators per tile
};

template <int TM, int TN, int WM, int WN, int WK, int BKC, int NBUF = 2>
__global__ __launch_bounds__(64 * WM * WN * WK) void gemm_nt_kernel(GemmBatch G) {
    using GG = GemmGeom<TM, TN, WM, WN, WK, BKC, NBUF>;
    constexpr int NT = GG::NT;
    constexpr int BM = GG::BM, BN = GG::BN, LDT = GG::LDT, C4 = GG::C4, STAGE = GG::STAGE;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const GemmProblem& P = G.p[blockIdx.y];
    if ((int)blockIdx.z >= max(P.nbatch, 1)) return;
    // problem fields once into registers (re-reading the kernarg segment inside the k-loop costs a scalar-load
    // round trip per use)
    const float* __restrict__ pA = P.A + (size_t)blockIdx.z * P.sA;
    const float* __restrict__ pA2 = P.A2;
    const float* __restrict__ pW = P.W + (size_t)blockIdx.z * P.sW;
    const int rows = P.rows, ncols = P.ncols, K = P.K, K1 = pA2 ? P.K1 : P.K, lda = P.lda, lda2 = P.lda2;
    const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (rows + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wk = w % WK, wn = (w / WK) % WN, wm = w / (WK * WN);
    const int nchunks = (K + BKC - 1) / BKC;

    // per-thread staging slots: (row, k-offset) are loop invariant; out-of-range rows are clamped to a valid
    // row and zeroed by a select, so the loads carry no branches
    float4 ra[GG::A_SLOTS], rb[GG::B_SLOTS];
    const float* a1p[GG::A_SLOTS];
    const float* a2p[GG::A_SLOTS];
    const float* bp[GG::B_SLOTS];
    int akc[GG::A_SLOTS], bkc[GG::B_SLOTS];
    bool aok[GG::A_SLOTS], bok[GG::B_SLOTS];
#pragma unroll
    for (int s = 0; s < GG::A_SLOTS; ++s) {
        const int slot = t + s * NT, r = slot / C4;
        akc[s] = 4 * (slot % C4);
        aok[s] = slot < BM * C4 && row0 + r < rows;
        const int rc = min(row0 + r, rows - 1);
        a1p[s] = pA + (size_t)rc * lda;
        a2p[s] = pA2 ? pA2 + (size_t)rc * lda2 - K1 : a1p[s];
    }
#pragma unroll
    for (int s = 0; s < GG::B_SLOTS; ++s) {
        const int slot = t + s * NT, r = slot / C4;
        bkc[s] = 4 * (slot % C4);
        bok[s] = slot < BN * C4 && col0 + r < ncols;
        bp[s] = pW + (size_t)min(col0 + r, ncols - 1) * K;
    }
    auto load_chunk = [&](int ch) {
        const int k0 = ch * BKC;
#pragma unroll
        for (int s = 0; s < GG::A_SLOTS; ++s) {
            const int k = k0 + akc[s];
            const int kc = min(k, K - 4);
            const float* src = (kc < K1 ? a1p[s] : a2p[s]) + kc;
            ra[s] = *reinterpret_cast<const float4*>(src);      // zeroing of invalid slots happens at store time:
        }                                                       // touching the data here would wait for the load
#pragma unroll
        for (int s = 0; s < GG::B_SLOTS; ++s) {
            const int k = k0 + bkc[s];
            rb[s] = *reinterpret_cast<const float4*>(bp[s] + min(k, K - 4));
        }
    };
    auto store_chunk = [&](int ch) {
        float* As = smem + (NBUF == 2 ? (ch & 1) : 0) * STAGE;
        float* Bs = As + BM * LDT;
        const int k0 = ch * BKC;
#pragma unroll
        for (int s = 0; s < GG::A_SLOTS; ++s) {
            const int slot = t + s * NT;
            const bool ok = aok[s] && k0 + akc[s] < K;
            const float4 v = ra[s];
            if (slot < BM * C4)
                *reinterpret_cast<float4*>(As + (slot / C4) * LDT + akc[s]) =
                    make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
#pragma unroll
        for (int s = 0; s < GG::B_SLOTS; ++s) {
            const int slot = t + s * NT;
            const bool ok = bok[s] && k0 + bkc[s] < K;
            const float4 v = rb[s];
            if (slot < BN * C4)
                *reinterpret_cast<float4*>(Bs + (slot / C4) * LDT + bkc[s]) =
                    make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };

    f32x16 acc[TM][TN][GG::NACC];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int n = 0; n < GG::NACC; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][n][r] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const int h = lane >> 5, l31 = lane & 31;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) load_chunk(ch + 1);
        const float* As = smem + (NBUF == 2 ? (ch & 1) : 0) * STAGE + (wm * TM * 32 + l31) * LDT + wk * GG::GROUPS * 8 + 4 * h;
        const float* Bs = smem + (NBUF == 2 ? (ch & 1) : 0) * STAGE + BM * LDT + (wn * TN * 32 + l31) * LDT + wk * GG::GROUPS * 8 + 4 * h;
        // fragments of group g+1 are fetched while the MFMAs of group g issue
        float4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const float4*>(As + i * 32 * LDT);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDT);
#pragma unroll
        for (int g = 0; g < GG::GROUPS; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < GG::GROUPS) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = *reinterpret_cast<const float4*>(As + i * 32 * LDT + 8 * (g + 1));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDT + 8 * (g + 1));
            }
            // consecutive MFMAs go to DIFFERENT accumulators (tiles, or the even/odd-group pair of a single
            // tile): a dependent accumulate chain alone does not keep the 64-cycle pipe full
            constexpr int NA = GG::NACC;
            const int q = (NA == 2) ? (g & 1) : 0;
#define DR_MFMA_STEP(E)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)              \
        acc[i][j][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].E, b[cur][j].E, acc[i][j][q], 0, 0, 0);
            DR_MFMA_STEP(x)
            DR_MFMA_STEP(y)
            DR_MFMA_STEP(z)
            DR_MFMA_STEP(w)
#undef DR_MFMA_STEP
        }
        if (NBUF == 1) __syncthreads();          // single buffer: everyone is done reading before it is overwritten
        if (ch + 1 < nchunks) store_chunk(ch + 1);
        __syncthreads();
    }

    if (GG::NACC == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][0][r] += acc[i][j][GG::NACC - 1][r];
    }
    if (WK > 1) {
        // reduce the WK partial accumulators of each (wm, wn) through LDS
        float* red = smem;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((w * TM + i) * TN + j) * 16 + r) * 64 + lane] = acc[i][j][0][r];
        __syncthreads();
        if (wk != 0) return;
#pragma unroll
        for (int o = 1; o < WK; ++o)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][0][r] += red[((((w + o) * TM + i) * TN + j) * 16 + r) * 64 + lane];
    }

    gemm_epilogue<TM, TN>(P, [&](int i, int j) -> const f32x16& { return acc[i][j][0]; }, rows, ncols, row0, col0, wm, wn, lane,
                          (size_t)blockIdx.z * P.sO);
}

template <int TM, int TN, int WM, int WN, int WK, int BKC, int NBUF = 2>
static int configure_cfg() {
    using GG = GemmGeom<TM, TN, WM, WN, WK, BKC, NBUF>;
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<TM, TN, WM, WN, WK, BKC, NBUF>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GG::SMEM_FLOATS * sizeof(float))));
    return DR_OK;
}

template <int TM, int TN, int WM, int WN, int WK, int BKC, int NBUF = 2>
static int launch_cfg(const GemmBatch& g, hipStream_t st) {
    using GG = GemmGeom<TM, TN, WM, WN, WK, BKC, NBUF>;
    int maxt = 0, maxb = 1;
    for (int i = 0; i < g.n; ++i) {
        const int tl = ((g.p[i].rows + GG::BM - 1) / GG::BM) * ((g.p[i].ncols + GG::BN - 1) / GG::BN);
        maxt = tl > maxt ? tl : maxt;
        maxb = g.p[i].nbatch > maxb ? g.p[i].nbatch : maxb;
    }
    if (maxt == 0) return DR_OK;
    double flops = 0;
    for (int i = 0; i < g.n; ++i) flops += 2.0 * g.p[i].rows * g.p[i].ncols * g.p[i].K * (g.p[i].nbatch > 1 ? g.p[i].nbatch : 1);
    ProfScope ps(PK_GEMM, flops, st);
    hipLaunchKernelGGL((gemm_nt_kernel<TM, TN, WM, WN, WK, BKC, NBUF>), dim3(maxt, g.n, maxb), dim3(GG::NT), GG::SMEM_FLOATS * sizeof(float), st, g);
    DR_LAUNCH_CHECK();
    return DR_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Latency form for the single-pair case (a few hundred rows: every GEMM of the loop is one short wave of workgroups
// whose time is load latency, not arithmetic).  One 32 x 32 output tile per workgroup, 8 waves, the k range dealt to
// the waves in groups of 8 (wave w: groups w, w + 8, ..); a lane's MFMA fragments are float4s of "its" row, so they
// are loaded STRAIGHT from global memory into registers -- all of a wave's loads are in flight at once, there is no
// LDS staging, no k loop with a barrier per chunk -- then 4 f32-input MFMAs per group, and one LDS pass adds the NW
// partial tiles (NW = 8 or 16 waves by the length of k; a wave reduces and stores 16 / NW registers of the tile).
template <int NW, int MAXG>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 4))) void gemm_nt_direct_kernel(GemmBatch G) {
    __shared__ float red[NW * 16 * 64];
    const GemmProblem& P = G.p[blockIdx.y];
    if ((int)blockIdx.z >= max(P.nbatch, 1)) return;
    const float* __restrict__ pA = P.A + (size_t)blockIdx.z * P.sA;
    const float* __restrict__ pA2 = P.A2;
    const float* __restrict__ pW = P.W + (size_t)blockIdx.z * P.sW;
    const int rows = P.rows, ncols = P.ncols, K = P.K, K1 = pA2 ? P.K1 : P.K;
    const int tiles_n = (ncols + 31) / 32, tiles_m = (rows + 31) / 32;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int row0 = tm * 32, col0 = tn * 32;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, l31 = lane & 31;
    const int ngroups = (K + 7) / 8;

    const int ar = min(row0 + l31, rows - 1), bc = min(col0 + l31, ncols - 1);
    const float* a1 = pA + (size_t)ar * P.lda;
    const float* a2 = pA2 ? pA2 + (size_t)ar * P.lda2 - K1 : a1;
    const float* bw = pW + (size_t)bc * K;
    float4 fa[MAXG], fb[MAXG];
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const int k = 8 * (w + NW * i) + 4 * h;                 // K % 4 == 0: a float4 is all inside or all outside
        const int kc = min(k, K - 4);
        fa[i] = *reinterpret_cast<const float4*>((kc < K1 ? a1 : a2) + kc);
        fb[i] = *reinterpret_cast<const float4*>(bw + kc);
    }
    __builtin_amdgcn_sched_barrier(0);                          // every load is issued before the first MFMA waits for one
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        const bool ok = 8 * (w + NW * i) + 4 * h < K;           // groups past the end (and the k tail of the last one) add zeros
        float4 a = fa[i];
        if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = fb[i];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[1], 0, 0, 0);
    }
    (void)ngroups;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[0][r] + acc[1][r];
    __syncthreads();
    // wave w finishes registers (16 / NW) w ..: rows (r & 3) + 8 (r >> 2) + 4 h, column l31
    const int halfC = P.rot_C >> 1;
    const int col = col0 + l31;
    const bool col_ok = col < ncols;
    const int ridx = (P.epi & EPI_ROTARY) ? (col % P.rot_C) >> 1 : 0;
    float* __restrict__ outp = P.out + (size_t)blockIdx.z * P.sO;
#pragma unroll
    for (int e = 0; e < 16 / NW; ++e) {
        const int r = (16 / NW) * w + e;
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < NW; ++o) v += red[(o * 16 + r) * 64 + lane];
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (P.epi & EPI_ROTARY) {
            const float other = __shfl_xor(v, 1);
            if (row < rows && col_ok) {
                const float c = P.cosT[(size_t)row * halfC + ridx], sn = P.sinT[(size_t)row * halfC + ridx];
                const float sw = (col & 1) ? other : -other;
                v = __fadd_rn(__fmul_rn(v, c), __fmul_rn(sw, sn));
            }
        }
        if (P.bias && col_ok) v += P.bias[col];
        if (P.epi & EPI_RELU) v = fmaxf(v, 0.f);
        v *= P.scale;
        if (row < rows && col_ok) {
            if (P.addend) v += P.addend[(size_t)row * P.ldo + col];
            outp[(size_t)row * P.ldo + col] = v;
        }
    }
}

template <int NW, int MAXG>
static int launch_direct(const GemmBatch& g, hipStream_t st) {
    int maxt = 0, maxb = 1;
    double flops = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tl = ((g.p[i].rows + 31) / 32) * ((g.p[i].ncols + 31) / 32);
        maxt = tl > maxt ? tl : maxt;
        maxb = g.p[i].nbatch > maxb ? g.p[i].nbatch : maxb;
        flops += 2.0 * g.p[i].rows * g.p[i].ncols * g.p[i].K * (g.p[i].nbatch > 1 ? g.p[i].nbatch : 1);
    }
    if (maxt == 0) return DR_OK;
    ProfScope ps(PK_GEMM, flops, st);
    hipLaunchKernelGGL((gemm_nt_direct_kernel<NW, MAXG>), dim3(maxt, g.n, maxb), dim3(64 * NW), 0, st, g);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

//                 TM TN WM WN WK BKC
#define CFG_SMALL  1, 1, 1, 1, 4, 128    /*  32 x  32 tile, k split over the 4 waves, deep chunks (latency-bound sizes) */
#define CFG_MEDIUM 1, 1, 2, 2, 1, 64     /*  64 x  64 tile                                                            */
#define CFG_LARGE  2, 1, 2, 2, 1, 32     /* 128 x  64 tile, 64 x 32 per wave                                          */
#define CFG_M1B    1, 1, 2, 2, 1, 32, 1  /*  64 x  64 tile, single LDS buffer: 18 KB -> 8 workgroups per CU            */

int gemm_configure() {
    int rc = configure_cfg<CFG_SMALL>();
    if (rc == DR_OK) rc = configure_cfg<CFG_MEDIUM>();
    if (rc == DR_OK) rc = configure_cfg<CFG_LARGE>();
    if (rc == DR_OK) rc = configure_cfg<CFG_M1B>();
    return rc;
}

static int g_force_cfg = -1;   // tools / tests: force a configuration (0, 1, 2, 9: LDS-staged f32-MFMA tiles; 11, 12: latency form)
void gemm_force_config(int c) { g_force_cfg = c; }

int launch_gemm(const GemmBatch& g, hipStream_t st) {
    if (g.n < 1 || g.n > 4) return DR_EINVAL;
    long nM = 0;                    // 64 x 64 tiles of the launch
    for (int i = 0; i < g.n; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.K % 4 || p.lda % 4 || (p.A2 && (p.K1 % 4 || p.lda2 % 4))) return DR_ENOSUP;
        if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.A2) & 15) return DR_ENOSUP;
        nM += (long)((p.rows + 63) / 64) * ((p.ncols + 63) / 64) * (p.nbatch > 1 ? p.nbatch : 1);
    }
    // latency form: few tiles (the whole launch is one short wave of workgroups) and a k range that fits the registers
    int maxK = 0;
    long n32 = 0;
    for (int i = 0; i < g.n; ++i) {
        maxK = g.p[i].K > maxK ? g.p[i].K : maxK;
        n32 += (long)((g.p[i].rows + 31) / 32) * ((g.p[i].ncols + 31) / 32) * (g.p[i].nbatch > 1 ? g.p[i].nbatch : 1);
    }
    const int direct_max = env_knob("DR_GEMM_DIRECT_MAX", 2048);   // (1193 rows x 1296 columns = 1558 tiles: 21 us against 36 us for the LDS-staged tiles)
    if (g_force_cfg < 0 && n32 <= direct_max && maxK <= 16 * 8 * 7) {
        // latency form or the 64 x 64 staged tiles?  Two fitted costs in us (tools/gemm_small.py, SHAPES=mid, 14 shapes between 1 024 x 256 x 256
        // and 2 048 x 864 x 864: the rule picks the faster kernel on every one of them):
        //   staged: 5 + rounds (0.0244 K), rounds = ceil(64 x 64 tiles / CUs): a workgroup per CU and round, MFMA-bound in it; it moves
        //           half the L2 bytes per output tile
        //   latency form: K <= 448 (8 waves):  2.5 + 0.0145 tiles;   K <= 896 (16 waves, 2 workgroups per CU): 11 up to 512 tiles, then 10 + 0.026 tiles
        // (2D-3D loop, 3 072 x 256 x 256: 13.5 -> 10.8 us per launch; 2 048 x 864 x 864: 55 -> 42; a single pair's 512 or 1 193 rows keep the latency form)
        const int n_cu = device_cu_count();
        const double t_staged = 5.0 + (double)((nM + n_cu - 1) / n_cu) * 0.0244 * maxK;
        const double t_direct = maxK <= 8 * 8 * 7 ? 2.5 + 0.0145 * (double)n32 : (n32 <= 512 ? 11.0 : 10.0 + 0.026 * (double)n32);
        if (t_direct <= t_staged || nM < 128) return maxK <= 8 * 8 * 7 ? launch_direct<8, 7>(g, st) : launch_direct<16, 7>(g, st);
    }
    int cfg = nM >= 128 ? 9 : 0;     // 9 = 64 x 64 tiles with a single LDS buffer (18 KB -> 8 workgroups per CU): best of
                                     // every f32-MFMA configuration measured on the loop's shapes (tools/gemm_bench.py)
    const int env_cfg = env_knob("DR_GEMM_CFG", -1);   // tools/: tile experiments
    if (env_cfg >= 0 && cfg == 9) cfg = env_cfg;
    if (g_force_cfg >= 0) cfg = g_force_cfg;
    if (cfg == 11) return launch_direct<8, 7>(g, st);
    if (cfg == 12) return launch_direct<16, 7>(g, st);
    if (cfg == 9) return launch_cfg<CFG_M1B>(g, st);
    if (cfg == 2) return launch_cfg<CFG_LARGE>(g, st);
    if (cfg == 1) return launch_cfg<CFG_MEDIUM>(g, st);
    if (cfg == 0) return launch_cfg<CFG_SMALL>(g, st);
    return DR_EINVAL;
}

}  // namespace dr
