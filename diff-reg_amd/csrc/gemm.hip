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
// Split-operand GEMM: fp32 result from bf16 MFMAs.  Every fp32 operand x is written (round-to-nearest at each
// level) as hi + mid + lo, three bf16 numbers, exact to ~2^-27 |x|; the product keeps the six terms down to
// 2^-18 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) and accumulates them in fp32 on v_mfma_f32_32x32x16_bf16.
// bf16 x bf16 products are exact in fp32, so the dropped terms (<= 2^-26 relative) are below the rounding of the
// fp32 accumulation itself: measured against an fp64 GEMM the result is as close as the f32-input MFMA kernel's
// (tests/test_ops_gpu.py::test_gemm_split_accuracy).  Six bf16 MFMAs (32 cycles each, K = 16) replace eight
// f32 MFMAs (64 cycles each, K = 2): 2.67x the f32-input MFMA rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const f32x2 f = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 r2 = {r.x - __uint_as_float(mid << 16), r.y - __uint_as_float(mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

__device__ __forceinline__ void split8_store(char* dst, int plane_bytes, const float4& u, const float4& v, bool ok) {
    uint4 hi, mid, lo;
    split_pair(ok ? u.x : 0.f, ok ? u.y : 0.f, hi.x, mid.x, lo.x);
    split_pair(ok ? u.z : 0.f, ok ? u.w : 0.f, hi.y, mid.y, lo.y);
    split_pair(ok ? v.x : 0.f, ok ? v.y : 0.f, hi.z, mid.z, lo.z);
    split_pair(ok ? v.z : 0.f, ok ? v.w : 0.f, hi.w, mid.w, lo.w);
    *reinterpret_cast<uint4*>(dst) = hi;
    *reinterpret_cast<uint4*>(dst + plane_bytes) = mid;
    *reinterpret_cast<uint4*>(dst + 2 * plane_bytes) = lo;
}


// ---------------------------------------------------------------------------------------------------------
// Wide split-operand kernel: the layer GEMMs (rows = thousands of tokens, ncols = K = C or 2C).
// What bounds the 64..128-column tiles above is not the MFMA but the staging: every column tile re-fetches and
// re-splits its A rows (7 times for C = 432) and splits the same weights again in every row tile.  Here
//   * the weights are split ONCE per call into a packed image (pack_weights_kernel) that IS the LDS image of a
//     224-column B tile, k-chunk by k-chunk: [K/16][ncols/224][224 rows][3 planes][16 k] bf16 with the 16-byte
//     row pad included (112-byte rows = 7 16-B slots, odd -> conflict-free ds_read_b128) and zero-filled past
//     ncols / K.  A B tile chunk is then 25 KB contiguous in global memory and is copied by LDS-DMA
//     (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPRs, nothing for the compiler to re-schedule);
//   * a workgroup spans 224 columns (two column tiles for C = 432), its 4 waves stacked on rows, each wave
//     holding a 32 x 224 strip of accumulators and streaming the B fragments through registers, so an A element
//     is fetched and split only ceil(ncols / 224) times.
// Rows past the end are clamped (their results are never stored); k past the end multiplies packed zeros.
// Per k-chunk: DMA of the next B chunk and the global loads of the A chunk after next are issued first; the
// split of the next A chunk into LDS sits between the MFMA groups of the column tiles; one barrier.
struct WideGeom {
    static constexpr int TN = 7, NWV = 4, BK = 16;
    static constexpr int NT = 64 * NWV, BM = 32 * NWV, BN = 32 * TN;
    static constexpr int PL = BK * 2, ROWB = 3 * PL + 16;
    static constexpr int A_BYTES = BM * ROWB;                    // 14,336
    static constexpr int B_IMG = ((BN * ROWB + 1023) / 1024) * 1024;   // 25,600: whole 1 KB DMA instructions
    static constexpr int B_DMAS = B_IMG / 1024;                  // 25 wave-instructions per chunk
    static constexpr int STAGE = A_BYTES + B_IMG, SMEM = 2 * STAGE;    // 79,872 B -> two workgroups per CU
};

__device__ long long g_gemm_stamps[256];
#define GEMM_STAMP(i) do { if (ABL == 9 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (i) < 256) g_gemm_stamps[i] = wall_clock64(); } while (0)
int read_gemm_stamps(long long* h_out) {
    DR_HIP_CHECK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gemm_stamps), sizeof(long long) * 256));
    return DR_OK;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_wide_kernel(GemmBatch G) {
    using GG = WideGeom;
    constexpr int TN = GG::TN, BM = GG::BM, BN = GG::BN, ROWB = GG::ROWB, PL = GG::PL, BK = GG::BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const GemmProblem& P = G.p[blockIdx.y];
    const float* __restrict__ pA = P.A;
    const float* __restrict__ pA2 = P.A2;
    const int rows = P.rows, ncols = P.ncols, K = P.K, K1 = pA2 ? P.K1 : P.K, lda = P.lda, lda2 = P.lda2;
    const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (rows + BM - 1) / BM;
    // workgroup id -> tile, XCD-aware: ids are dealt round-robin to the 8 XCDs (each with its own L2), so the column
    // tiles of one row block get ids 8 apart: they share an L2 and the A rows cross the fabric once, not tiles_n times
    const int grp = blockIdx.x / (8 * tiles_n), rem = blockIdx.x % (8 * tiles_n);
    const int tm = grp * 8 + (rem & 7), tn = rem >> 3;
    if (tm >= tiles_m) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nchunks = (K + BK - 1) / BK;

    // A: thread t stages 8 consecutive k of row t / 2 (one float4 pair per chunk); two register sets alternate
    const int ar = t >> 1, akc = 8 * (t & 1), alds = ar * ROWB + akc * 2;
    const float* a1p = pA + (size_t)min(row0 + ar, rows - 1) * lda;
    const float* a2p = pA2 ? pA2 + (size_t)min(row0 + ar, rows - 1) * lda2 - K1 : a1p;
    float4 ra[2][2];
    auto load_a = [&](int ch, float4 (&dst)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kc = min(ch * BK + akc + 4 * q, K - 4);
            dst[q] = *reinterpret_cast<const float4*>((kc < K1 ? a1p : a2p) + kc);
        }
    };
    // B: wave w issues the DMA instructions w, w + 4, ... of the 25 that copy one packed tile chunk
    const char* bsrc = reinterpret_cast<const char*>(P.Wsplit) + (size_t)tn * GG::B_IMG + lane * 16;
    const size_t bstep = (size_t)tiles_n * GG::B_IMG;
    auto dma_b = [&](int ch) {
        const char* src = bsrc + (size_t)ch * bstep;
        char* dst = lds + (ch & 1) * GG::STAGE + GG::A_BYTES;
#pragma unroll
        for (int i = 0; i < (GG::B_DMAS + 3) / 4; ++i) {
            const int ins = w + 4 * i;
            if (ins < GG::B_DMAS)
                __builtin_amdgcn_global_load_lds((glb_void*)(src + ins * 1024), (lds_void*)(dst + ins * 1024), 16, 0, 0);
        }
    };
    // The split planes are written with asm ds_write_b128: hipcc orders every LDS store it knows about behind the
    // LDS-DMA in flight (s_waitcnt vmcnt(0)), which would also wait for the A loads just issued for chunk ch + 2.
    // The DMA and these stores touch disjoint bytes of the stage; lgkmcnt(0) before the barrier retires them.
    const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
    auto store_a = [&](int ch, const float4 (&src)[2]) {
        const unsigned d = lds_base + (ch & 1) * GG::STAGE + alds;
        uint4 hi, mid, lo;
        if (ABL == 1) {
            hi = __builtin_bit_cast(uint4, src[0]); mid = __builtin_bit_cast(uint4, src[1]); lo = hi;
        } else {
            split_pair(src[0].x, src[0].y, hi.x, mid.x, lo.x);
            split_pair(src[0].z, src[0].w, hi.y, mid.y, lo.y);
            split_pair(src[1].x, src[1].y, hi.z, mid.z, lo.z);
            split_pair(src[1].z, src[1].w, hi.w, mid.w, lo.w);
        }
        const u32x4 vh = {hi.x, hi.y, hi.z, hi.w}, vm = {mid.x, mid.y, mid.z, mid.w}, vl = {lo.x, lo.y, lo.z, lo.w};
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:32\n\tds_write_b128 %0, %3 offset:64"
                     :: "v"(d), "v"(vh), "v"(vm), "v"(vl) : "memory");
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;

    // One k-chunk; cur = register set holding A(ch + 1), nxt = the set A(ch + 2) is fetched into.
    // A column tile is 6 dependent MFMAs (32 cycles each); the staging of the next chunk is cut into pieces that are
    // issued in the gaps between them, pinned there by sched_barrier (left alone hipcc gathers them after the MFMAs,
    // where nothing hides them):  gap G = 5 j + g  (tile j, after its MFMA g)
    //   G 0..6   one B DMA instruction each (wave w: instructions w, w + 4, ..)      -> landed long before the barrier
    //   G 7, 8   the two float4 of A(ch + 2)
    //   G 15..18 split of A(ch + 1), one pair of planes' worth per gap;  G 19 its three ds_write_b128
    const int wu = __builtin_amdgcn_readfirstlane(w);
    auto chunk = [&](int ch, float4 (&cur)[2], float4 (&nxt)[2]) __attribute__((always_inline)) {
        const bool has_next = ch + 1 < nchunks;
        const bool do_load = has_next && ABL != 2;
        GEMM_STAMP(8 + 8 * ch);
        const unsigned sbase = lds_base + (ch & 1) * GG::STAGE;
        const unsigned Ab = sbase + (w * 32 + l31) * ROWB + 16 * h;
        const unsigned Bb = sbase + GG::A_BYTES + l31 * ROWB + 16 * h;
        const char* dsrc = bsrc + (size_t)(ch + 1) * bstep;
        char* ddst = lds + ((ch + 1) & 1) * GG::STAGE + GG::A_BYTES;
        const unsigned adst = lds_base + ((ch + 1) & 1) * GG::STAGE + alds;
        uint4 hi, mid, lo;
        auto gap = [&](int G) __attribute__((always_inline)) {
            if (G < 7) {
                const int ins = wu + 4 * G;
                if (do_load && ins < GG::B_DMAS)
                    __builtin_amdgcn_global_load_lds((glb_void*)(dsrc + ins * 1024), (lds_void*)(ddst + ins * 1024), 16, 0, 0);
            } else if (G < 9) {
                if (do_load) {
                    const int kc = min((ch + 2) * BK + akc + 4 * (G - 7), K - 4);
                    nxt[G - 7] = *reinterpret_cast<const float4*>((kc < K1 ? a1p : a2p) + kc);
                }
            } else if (G >= 15 && G < 19 && has_next) {
                if (ABL == 1) {
                    if (G == 15) { hi = __builtin_bit_cast(uint4, cur[0]); mid = __builtin_bit_cast(uint4, cur[1]); lo = hi; }
                } else if (G == 15) split_pair(cur[0].x, cur[0].y, hi.x, mid.x, lo.x);
                else if (G == 16) split_pair(cur[0].z, cur[0].w, hi.y, mid.y, lo.y);
                else if (G == 17) split_pair(cur[1].x, cur[1].y, hi.z, mid.z, lo.z);
                else split_pair(cur[1].z, cur[1].w, hi.w, mid.w, lo.w);
            } else if (G == 19 && has_next) {
                const u32x4 vh = {hi.x, hi.y, hi.z, hi.w}, vm = {mid.x, mid.y, mid.z, mid.w}, vl = {lo.x, lo.y, lo.z, lo.w};
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:32\n\tds_write_b128 %0, %3 offset:64"
                             :: "v"(adst), "v"(vh), "v"(vm), "v"(vl) : "memory");
            }
        };
        // Fragment reads are asm ds_read_b128 with hand-counted waits: hipcc's own bookkeeping answers the first use
        // of tile j's fragments with lgkmcnt(0), which also waits for the prefetch of tile j + 1 just issued
        // (LDS operations retire in order, so "all but the newest 3" is the wait that is needed).
        u32x4 a[3], b[2][3];
#define DR_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#pragma unroll
        for (int p = 0; p < 3; ++p) DR_LDS_READ(a[p], Ab, p * PL);
#pragma unroll
        for (int p = 0; p < 3; ++p) DR_LDS_READ(b[0][p], Bb, p * PL);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cb = j & 1;
            if (j + 1 < TN) {                                   // fragments of tile j + 1 fly while tile j multiplies
#pragma unroll
                for (int p = 0; p < 3; ++p) DR_LDS_READ(b[cb ^ 1][p], Bb, (j + 1) * 32 * ROWB + p * PL);
                // the ds_writes of gap 19 sit between the reads of tile 5 and those of tile 4 in the LDS queue
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);                  // MFMAs must not be hoisted above the wait (guide rule 18)
            if (j == 0) GEMM_STAMP(8 + 8 * ch + 2);
            if (j == 4) GEMM_STAMP(8 + 8 * ch + 4);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[0]), a1 = __builtin_bit_cast(bf16x8, a[1]), a2 = __builtin_bit_cast(bf16x8, a[2]);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[cb][0]), b1 = __builtin_bit_cast(bf16x8, b[cb][1]), b2 = __builtin_bit_cast(bf16x8, b[cb][2]);
#define DR_MFMA_GAP(X, Y, g)                                                              \
    if (ABL != 3) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Y, acc[j], 0, 0, 0); \
    else asm volatile("" ::"v"(X), "v"(Y));                                                \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (g < 5) { gap(5 * j + g); __builtin_amdgcn_sched_barrier(0); }
            // smallest terms first
            DR_MFMA_GAP(a2, b0, 0)
            DR_MFMA_GAP(a0, b2, 1)
            DR_MFMA_GAP(a1, b1, 2)
            DR_MFMA_GAP(a1, b0, 3)
            DR_MFMA_GAP(a0, b1, 4)
            DR_MFMA_GAP(a0, b0, 5)
#undef DR_MFMA_GAP
            if (j == 3) GEMM_STAMP(8 + 8 * ch + 3);
        }
#undef DR_LDS_READ
        GEMM_STAMP(8 + 8 * ch + 5);
    };

    GEMM_STAMP(0);
    dma_b(0);
    load_a(0, ra[0]);
    load_a(1, ra[1]);
    store_a(0, ra[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += 2) {
        chunk(c0, ra[1], ra[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        GEMM_STAMP(8 + 8 * c0 + 6);
        if (c0 + 1 < nchunks) {
            chunk(c0 + 1, ra[0], ra[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            GEMM_STAMP(8 + 8 * c0 + 14);
        }
    }
    GEMM_STAMP(1);
    // Epilogue through LDS (the stage buffers are free after the last barrier): the MFMA result layout has a lane's
    // 16 values in 16 different rows, which stored directly is 112 dword stores per lane in 128-byte row pieces with
    // per-element address arithmetic.  Each wave transposes its 32 x 224 strip in two passes (4 then 3 column tiles)
    // through a private [32][136] float region, and every lane then owns float4s of whole rows: 512-byte row
    // segments, the rotary pair (2k, 2k+1) inside one float4, cos/sin as one float2, addend as a float4.
    {
        constexpr int EST = 136;                                // row stride (floats): rows 4 apart land 32 banks apart
        float* const ep = reinterpret_cast<float*>(lds) + w * 32 * EST;
        const int epi = P.epi, halfC = P.rot_C >> 1, rotC = P.rot_C, ldo = P.ldo;
        const float scale = P.scale;
        const float* __restrict__ bias = P.bias;
        const float* __restrict__ addend = P.addend;
        const float* __restrict__ cosT = P.cosT;
        const float* __restrict__ sinT = P.sinT;
        float* __restrict__ outp = P.out;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int j0 = pass * 4, nt = pass ? TN - 4 : 4;    // column tiles of this pass
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj >= nt) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[((r & 3) + 8 * (r >> 2) + 4 * h) * EST + jj * 32 + l31] = acc[j0 + jj][r];
            }
            // the wave re-reads only its own region: no workgroup barrier, the LDS queue is in order per wave
            const int c4 = (lane & 31) * 4;                     // column offset inside the pass
            const int col = col0 + j0 * 32 + c4;
            const bool col_ok = c4 < nt * 32 && col < ncols;    // ncols % 4 == 0 (launch check)
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias && col_ok) bv = *reinterpret_cast<const float4*>(bias + col);
            const int ridx = (epi & EPI_ROTARY) ? (col % rotC) >> 1 : 0;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rl = it * 2 + (lane >> 5), row = row0 + w * 32 + rl;
                float4 v = *reinterpret_cast<const float4*>(ep + rl * EST + c4);
                if (row < rows && col_ok) {
                    if (epi & EPI_ROTARY) {
                        // x*cos + swap(x)*sin, swap(x)[2k] = -x[2k+1], swap(x)[2k+1] = x[2k]  (position_encoding.py:25-35)
                        const float2 c = *reinterpret_cast<const float2*>(cosT + (size_t)row * halfC + ridx);
                        const float2 sn = *reinterpret_cast<const float2*>(sinT + (size_t)row * halfC + ridx);
                        const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
                        v.x = __fadd_rn(__fmul_rn(x0, c.x), __fmul_rn(-x1, sn.x));
                        v.y = __fadd_rn(__fmul_rn(x1, c.x), __fmul_rn(x0, sn.x));
                        v.z = __fadd_rn(__fmul_rn(x2, c.y), __fmul_rn(-x3, sn.y));
                        v.w = __fadd_rn(__fmul_rn(x3, c.y), __fmul_rn(x2, sn.y));
                    }
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (epi & EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                    float* o = outp + (size_t)row * ldo + col;
                    if (addend) {
                        const float4 ad = *reinterpret_cast<const float4*>(addend + (size_t)row * ldo + col);
                        v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                    }
                    *reinterpret_cast<float4*>(o) = v;
                }
            }
        }
    }
    GEMM_STAMP(2);
}

// ---------------------------------------------------------------------------------------------------------
// The wide kernel on a TWO-plane fp16 split -- the default packed GEMM (DR_GEMM_F16X2=0 / dr_debug_gemm_f16x2(0) select the
// three-plane bf16 kernel above instead).
// x = hi + lo with hi = fp16(x), lo = fp16(x - hi) keeps 22 significand bits; the three products hi*hi, hi*lo, lo*hi are
// exact in fp32 and the dropped lo*lo is 2^-22 relative -- below the fp32 accumulation error of a K = 432 dot product
// (tools/split_error_study.py: the same error as a float32 matmul for operands of scale 1 .. 30) -- at HALF the MFMA work
// of the three-plane bf16 split.  fp16 has 5 exponent bits, so both operands are brought into its range by EXACT powers of
// two that the epilogue undoes: every row of A by 2^s_r with s_r = 14 - floor(log2(max |row|)) (a pre-pass of the workgroup
// over its 128 rows, while the first B chunk is in flight), every row of W (output column) by 2^s_c likewise at pack time.
// Elements within 2^-17 of their row's maximum then keep 22 significand bits; smaller ones are rounded to 2^-39 of the row
// maximum absolutely -- either way far inside the fp32 rounding of a dot product that contains the row's maximum.
// Same structure as gemm_nt_wide_kernel: 128 x 224 tile, 4 waves, B by LDS-DMA, A split on the way into LDS, one barrier
// per 16-deep k-chunk; rows of the LDS images are [hi 32 B | lo 32 B | pad 16 B] = 80 B (5 slots: odd, conflict-free).
struct WideGeom2 {
    static constexpr int TN = 7, NWV = 4, BK = 16;
    static constexpr int NT = 64 * NWV, BM = 32 * NWV, BN = 32 * TN;
    static constexpr int PL = BK * 2, ROWB = 2 * PL + 16;
    static constexpr int A_BYTES = BM * ROWB;                          // 10,240
    static constexpr int B_IMG = ((BN * ROWB + 1023) / 1024) * 1024;   // 18,432
    static constexpr int B_DMAS = B_IMG / 1024;                        // 18 wave-instructions per chunk
    static constexpr int STAGE = A_BYTES + B_IMG;                      // 28,672
    static constexpr int EPI_BYTES = NWV * 32 * 136 * 4;               // the epilogue's transposition regions: 69,632
    static constexpr int WORK = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
    static constexpr int SMEM = WORK + BM * 4;                         // + 2^-s_r of the 128 rows: two workgroups per CU
    static constexpr int TARGET_EXP = 14;                              // scaled row maxima land in [2^14, 2^15)
};
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 f = {x0, x1};
    const f16x2 h = __builtin_convertvector(f, f16x2);                 // round to nearest even
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 r = {x0 - hf.x, x1 - hf.y};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// exponent s with 2^14 <= m 2^s < 2^15 (0 for m = 0 / inf / nan), clamped so that 2^s and 2^-s are normal floats
__device__ __forceinline__ int f16_scale_exp(float m) {
    const unsigned bits = __float_as_uint(m);
    const int e = (int)((bits >> 23) & 0xff) - 127;
    const bool ok = m > 0.f && e < 128;
    return ok ? min(max(WideGeom2::TARGET_EXP - e, -100), 100) : 0;
}
__device__ __forceinline__ float pow2i(int s) { return __uint_as_float((unsigned)(127 + s) << 23); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_wide2_kernel(GemmBatch G) {
    using GG = WideGeom2;
    constexpr int TN = GG::TN, BM = GG::BM, BN = GG::BN, ROWB = GG::ROWB, PL = GG::PL, BK = GG::BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const GemmProblem& P = G.p[blockIdx.y];
    const float* __restrict__ pA = P.A;
    const float* __restrict__ pA2 = P.A2;
    const int rows = P.rows, ncols = P.ncols, K = P.K, K1 = pA2 ? P.K1 : P.K, lda = P.lda, lda2 = P.lda2;
    const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (rows + BM - 1) / BM;
    const int grp = blockIdx.x / (8 * tiles_n), rem = blockIdx.x % (8 * tiles_n);      // XCD-aware map (see the 3-plane kernel)
    const int tm = grp * 8 + (rem & 7), tn = rem >> 3;
    if (tm >= tiles_m) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nchunks = (K + BK - 1) / BK;

    const int ar = t >> 1, akc = 8 * (t & 1), alds = ar * ROWB + akc * 2;
    const float* a1p = pA + (size_t)min(row0 + ar, rows - 1) * lda;
    const float* a2p = pA2 ? pA2 + (size_t)min(row0 + ar, rows - 1) * lda2 - K1 : a1p;
    float4 ra[2][2];
    auto load_a = [&](int ch, float4 (&dst)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kc = min(ch * BK + akc + 4 * q, K - 4);
            dst[q] = *reinterpret_cast<const float4*>((kc < K1 ? a1p : a2p) + kc);
        }
    };
    const char* bsrc = reinterpret_cast<const char*>(P.Wsplit) + (size_t)tn * GG::B_IMG + lane * 16;
    const size_t bstep = (size_t)tiles_n * GG::B_IMG;
    auto dma_b = [&](int ch) {
        const char* src = bsrc + (size_t)ch * bstep;
        char* dst = lds + (ch & 1) * GG::STAGE + GG::A_BYTES;
#pragma unroll
        for (int i = 0; i < (GG::B_DMAS + 3) / 4; ++i) {
            const int ins = w + 4 * i;
            if (ins < GG::B_DMAS)
                __builtin_amdgcn_global_load_lds((glb_void*)(src + ins * 1024), (lds_void*)(dst + ins * 1024), 16, 0, 0);
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
    float* const s_rinv = reinterpret_cast<float*>(lds + GG::WORK);
    float sc = 1.f;                                              // 2^s_r of this thread's row (set by the pre-pass below)
    auto store_a = [&](int ch, const float4 (&src)[2]) {
        const unsigned d = lds_base + (ch & 1) * GG::STAGE + alds;
        uint4 hi, lo;
        split_pair_f16(src[0].x * sc, src[0].y * sc, hi.x, lo.x);
        split_pair_f16(src[0].z * sc, src[0].w * sc, hi.y, lo.y);
        split_pair_f16(src[1].x * sc, src[1].y * sc, hi.z, lo.z);
        split_pair_f16(src[1].z * sc, src[1].w * sc, hi.w, lo.w);
        const u32x4 vh = {hi.x, hi.y, hi.z, hi.w}, vl = {lo.x, lo.y, lo.z, lo.w};
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:32" :: "v"(d), "v"(vh), "v"(vl) : "memory");
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int h = lane >> 5, l31 = lane & 31;

    // One k-chunk.  A column tile is 3 dependent MFMAs; gap G = 3 j + g (tile j, after its MFMA g):
    //   G 0..4   one B DMA instruction each (wave w: instructions w, w + 4, ..)
    //   G 5, 6   the two float4 of A(ch + 2)
    //   G 9..12  split of A(ch + 1), one register pair per gap;  G 13 its two ds_write_b128
    const int wu = __builtin_amdgcn_readfirstlane(w);
    auto chunk = [&](int ch, float4 (&cur)[2], float4 (&nxt)[2]) __attribute__((always_inline)) {
        const bool has_next = ch + 1 < nchunks;
        const unsigned sbase = lds_base + (ch & 1) * GG::STAGE;
        const unsigned Ab = sbase + (w * 32 + l31) * ROWB + 16 * h;
        const unsigned Bb = sbase + GG::A_BYTES + l31 * ROWB + 16 * h;
        const char* dsrc = bsrc + (size_t)(ch + 1) * bstep;
        char* ddst = lds + ((ch + 1) & 1) * GG::STAGE + GG::A_BYTES;
        const unsigned adst = lds_base + ((ch + 1) & 1) * GG::STAGE + alds;
        uint4 hi, lo;
        auto gap = [&](int Gi) __attribute__((always_inline)) {
            if (Gi < 5) {
                const int ins = wu + 4 * Gi;
                if (has_next && ins < GG::B_DMAS)
                    __builtin_amdgcn_global_load_lds((glb_void*)(dsrc + ins * 1024), (lds_void*)(ddst + ins * 1024), 16, 0, 0);
            } else if (Gi < 7) {
                if (has_next) {
                    const int kc = min((ch + 2) * BK + akc + 4 * (Gi - 5), K - 4);
                    nxt[Gi - 5] = *reinterpret_cast<const float4*>((kc < K1 ? a1p : a2p) + kc);
                }
            } else if (Gi >= 9 && Gi < 13 && has_next) {
                if (Gi == 9) split_pair_f16(cur[0].x * sc, cur[0].y * sc, hi.x, lo.x);
                else if (Gi == 10) split_pair_f16(cur[0].z * sc, cur[0].w * sc, hi.y, lo.y);
                else if (Gi == 11) split_pair_f16(cur[1].x * sc, cur[1].y * sc, hi.z, lo.z);
                else split_pair_f16(cur[1].z * sc, cur[1].w * sc, hi.w, lo.w);
            } else if (Gi == 13 && has_next) {
                const u32x4 vh = {hi.x, hi.y, hi.z, hi.w}, vl = {lo.x, lo.y, lo.z, lo.w};
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:32" :: "v"(adst), "v"(vh), "v"(vl) : "memory");
            }
        };
        u32x4 a[2], b[2][2];
#define DR_LDS_READ2(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#pragma unroll
        for (int p = 0; p < 2; ++p) DR_LDS_READ2(a[p], Ab, p * PL);
#pragma unroll
        for (int p = 0; p < 2; ++p) DR_LDS_READ2(b[0][p], Bb, p * PL);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cb = j & 1;
            if (j + 1 < TN) {
#pragma unroll
                for (int p = 0; p < 2; ++p) DR_LDS_READ2(b[cb ^ 1][p], Bb, (j + 1) * 32 * ROWB + p * PL);
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");          // all but the two reads just issued
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
            const f16x8 b0 = __builtin_bit_cast(f16x8, b[cb][0]), b1 = __builtin_bit_cast(f16x8, b[cb][1]);
#define DR_MFMA_GAP2(X, Y, g)                                                   \
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X, Y, acc[j], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                          \
    gap(3 * j + g);                                                             \
    __builtin_amdgcn_sched_barrier(0);
            // smallest terms first
            DR_MFMA_GAP2(a1, b0, 0)
            DR_MFMA_GAP2(a0, b1, 1)
            DR_MFMA_GAP2(a0, b0, 2)
#undef DR_MFMA_GAP2
        }
#undef DR_LDS_READ2
    };

    dma_b(0);
    {   // max |row| -> the row's scale.  The producer of A usually left it behind (LayerNorm and this kernel's own epilogue
        // write row maxima); otherwise a pre-pass over this thread's half of its row (independent loads, all in flight; a
        // coalesced row-at-a-time sweep with a wave reduction per row measured slower: 89 vs 67 us at 32768 x 432 x 432).
        float mx = 0.f;
        if (P.amax) {
            const int row = min(row0 + ar, rows - 1);
            for (int q = 0; q < P.amax_parts; ++q) mx = fmaxf(mx, P.amax[(size_t)q * P.amax_stride + row]);
            if (P.amax2) mx = fmaxf(mx, P.amax2[row]);
        } else {
            for (int ch = 0; ch < nchunks; ++ch) {
                float4 v[2];
                load_a(ch, v);
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0].x), fabsf(v[0].y)), fmaxf(fabsf(v[0].z), fabsf(v[0].w))));
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[1].x), fabsf(v[1].y)), fmaxf(fabsf(v[1].z), fabsf(v[1].w))));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1));                   // the two threads of a row
        }
        const int s = f16_scale_exp(mx);
        sc = pow2i(s);
        if ((t & 1) == 0) s_rinv[ar] = pow2i(-s);
    }
    load_a(0, ra[0]);
    load_a(1, ra[1]);
    store_a(0, ra[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c0 = 0; c0 < nchunks; c0 += 2) {
        chunk(c0, ra[1], ra[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (c0 + 1 < nchunks) {
            chunk(c0 + 1, ra[0], ra[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // epilogue through LDS, as in the 3-plane kernel (the wave's 32 x 224 strip in two passes); the weight scale is undone first
    {
        constexpr int EST = 136;
        float* const ep = reinterpret_cast<float*>(lds) + w * 32 * EST;
        const int epi = P.epi, halfC = P.rot_C >> 1, rotC = P.rot_C, ldo = P.ldo;
        const float scale = P.scale;
        const float* __restrict__ cinv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(P.Wsplit) + (size_t)nchunks * bstep);
        const float* __restrict__ bias = P.bias;
        const float* __restrict__ addend = P.addend;
        const float* __restrict__ cosT = P.cosT;
        const float* __restrict__ sinT = P.sinT;
        float* __restrict__ outp = P.out;
        float omx[16];                                           // max |stored value| of row (2 it + lane / 32), this lane's columns
#pragma unroll
        for (int it = 0; it < 16; ++it) omx[it] = 0.f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int j0 = pass * 4, nt = pass ? TN - 4 : 4;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj >= nt) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[((r & 3) + 8 * (r >> 2) + 4 * h) * EST + jj * 32 + l31] = acc[j0 + jj][r];
            }
            const int c4 = (lane & 31) * 4;
            const int col = col0 + j0 * 32 + c4;
            const bool col_ok = c4 < nt * 32 && col < ncols;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cf = make_float4(1.f, 1.f, 1.f, 1.f);
            if (bias && col_ok) bv = *reinterpret_cast<const float4*>(bias + col);
            if (col_ok) cf = *reinterpret_cast<const float4*>(cinv + col);       // 2^-s_c of the four columns
            const int ridx = (epi & EPI_ROTARY) ? (col % rotC) >> 1 : 0;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int rl = it * 2 + (lane >> 5), row = row0 + w * 32 + rl;
                float4 v = *reinterpret_cast<const float4*>(ep + rl * EST + c4);
                if (row < rows && col_ok) {
                    const float rf = s_rinv[w * 32 + rl];        // undo the operand scales: exact powers of two
                    v.x = v.x * rf * cf.x; v.y = v.y * rf * cf.y; v.z = v.z * rf * cf.z; v.w = v.w * rf * cf.w;
                    if (epi & EPI_ROTARY) {
                        const float2 c = *reinterpret_cast<const float2*>(cosT + (size_t)row * halfC + ridx);
                        const float2 sn = *reinterpret_cast<const float2*>(sinT + (size_t)row * halfC + ridx);
                        const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
                        v.x = __fadd_rn(__fmul_rn(x0, c.x), __fmul_rn(-x1, sn.x));
                        v.y = __fadd_rn(__fmul_rn(x1, c.x), __fmul_rn(x0, sn.x));
                        v.z = __fadd_rn(__fmul_rn(x2, c.y), __fmul_rn(-x3, sn.y));
                        v.w = __fadd_rn(__fmul_rn(x3, c.y), __fmul_rn(x2, sn.y));
                    }
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (epi & EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                    float* o = outp + (size_t)row * ldo + col;
                    if (addend) {
                        const float4 ad = *reinterpret_cast<const float4*>(addend + (size_t)row * ldo + col);
                        v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                    }
                    *reinterpret_cast<float4*>(o) = v;
                    omx[it] = fmaxf(omx[it], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
            }
        }
        if (P.omax) {                                            // one partial maximum per row and 224-column tile
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                float m = omx[it];
#pragma unroll
                for (int d = 16; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));     // over the 32 lanes that share the row
                const int row = row0 + w * 32 + it * 2 + (lane >> 5);
                if ((lane & 31) == 0 && row < rows) P.omax[(size_t)tn * P.omax_stride + row] = m;
            }
        }
    }
}

// W [ncols][K] fp32 -> per output column c the scale 2^s_c (stored as 2^-s_c behind the image) ...
__global__ __launch_bounds__(256) void wcol_scale_kernel(const float* __restrict__ W, int ncols, int K, float* __restrict__ cinv) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= ncols) return;
    float mx = 0.f;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(W[(size_t)c * K + k]));
    mx = wave_max(mx);
    if (lane == 0) cinv[c] = pow2i(-f16_scale_exp(mx));
}
// ... and the two-plane fp16 image of W[c][:] * 2^s_c
__global__ void pack_weights_f16_kernel(const float* __restrict__ W, char* __restrict__ out, int ncols, int K,
                                        const float* __restrict__ cinv) {
    using GG = WideGeom2;
    const int nck = (K + 15) / 16, tiles_n = (ncols + GG::BN - 1) / GG::BN;
    const size_t n = (size_t)nck * tiles_n * GG::BN * 2, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int half = (int)(i & 1);
    size_t rest = i >> 1;
    const int r = (int)(rest % GG::BN); rest /= GG::BN;
    const int tn = (int)(rest % tiles_n), ch = (int)(rest / tiles_n);
    const int col = tn * GG::BN + r, k = ch * 16 + half * 8;
    const float sc = col < ncols ? 1.0f / cinv[col] : 1.f;            // (a power of two: exact)
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (col < ncols && k + e < K) ? W[(size_t)col * K + k + e] * sc : 0.f;
    uint4 hi, lo;
    split_pair_f16(x[0], x[1], hi.x, lo.x);
    split_pair_f16(x[2], x[3], hi.y, lo.y);
    split_pair_f16(x[4], x[5], hi.z, lo.z);
    split_pair_f16(x[6], x[7], hi.w, lo.w);
    char* d = out + ((size_t)ch * tiles_n + tn) * GG::B_IMG + (size_t)r * GG::ROWB + half * 16;
    *reinterpret_cast<uint4*>(d) = hi;
    *reinterpret_cast<uint4*>(d + GG::PL) = lo;
}

static int g_f16x2 = -1;         // -1 = environment (DR_GEMM_F16X2, default ON); set BEFORE the weights are packed
void gemm_force_f16x2(int on) { g_f16x2 = on; }
bool gemm_f16x2() {
    static const int v = env_knob("DR_GEMM_F16X2", 1);
    return (g_f16x2 >= 0 ? g_f16x2 : v) != 0;
}

// ---------------------------------------------------------------------------------------------------------
// 8-wave form of the wide kernel: same 128 x 224 workgroup tile, same LDS images and DMA, but the 7 column tiles
// of a 32-row strip are shared by two waves (4 + 3 tiles; wave w: strip w & 3, half w >> 2, so the two halves of a
// strip sit on the same SIMD and balance it).  A workgroup then has two waves per SIMD whose MFMA groups fill
// each other's staging / wait / barrier gaps even when only one workgroup fits the launch on a CU (the layer
// GEMMs of one batch are 128..512 tiles on 256 CUs), and with 64 accumulator registers per wave two workgroups
// (four waves per SIMD) still fit.
template <int SUB, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_wide8_kernel(GemmBatch G) {
    // SUB = 16-deep k-chunks per LDS stage (per barrier): 1 -> 80 KB, two workgroups per CU; 2 -> 160 KB, one
    // workgroup per CU with twice the time for a stage's DMA to land and half the barriers.
    using GG = WideGeom;
    constexpr int BM = GG::BM, BN = GG::BN, ROWB = GG::ROWB, PL = GG::PL, BK = GG::BK;
    constexpr int SSTAGE = SUB * GG::STAGE;                     // a stage = SUB x (A image | B image)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const GemmProblem& P = G.p[blockIdx.y];
    const float* __restrict__ pA = P.A;
    const float* __restrict__ pA2 = P.A2;
    const int rows = P.rows, ncols = P.ncols, K = P.K, K1 = pA2 ? P.K1 : P.K, lda = P.lda, lda2 = P.lda2;
    const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (rows + BM - 1) / BM;
    const int grp = blockIdx.x / (8 * tiles_n), rem = blockIdx.x % (8 * tiles_n);     // XCD-aware, as in the 4-wave kernel
    const int tm = grp * 8 + (rem & 7), tn = rem >> 3;
    if (tm >= tiles_m) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(w), wm = wu & 3, wn = wu >> 2;
    const int nchunks = (K + BK - 1) / BK, nstages = (nchunks + SUB - 1) / SUB;
    const int h = lane >> 5, l31 = lane & 31;

    // A: thread t stages 4 consecutive k of row t / 4 per 16-chunk (64 contiguous bytes per row per load)
    const int ar = t >> 2, akc = 4 * (t & 3), alds = ar * ROWB + akc * 2;
    const float* a1p = pA + (size_t)min(row0 + ar, rows - 1) * lda;
    const float* a2p = pA2 ? pA2 + (size_t)min(row0 + ar, rows - 1) * lda2 - K1 : a1p;
    float4 ra[2][SUB];
    auto load_a = [&](int ch, float4& dst) {
        const int kc = min(ch * BK + akc, K - 4);
        dst = *reinterpret_cast<const float4*>((kc < K1 ? a1p : a2p) + kc);
    };
    const char* bsrc = reinterpret_cast<const char*>(P.Wsplit) + (size_t)tn * GG::B_IMG + lane * 16;
    const size_t bstep = (size_t)tiles_n * GG::B_IMG;
    const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
    auto split4 = [&](const float4& v, uint2& hi, uint2& mid, uint2& lo) {
        if (ABL == 1) { hi = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); mid = hi; lo = hi; return; }
        split_pair(v.x, v.y, hi.x, mid.x, lo.x);
        split_pair(v.z, v.w, hi.y, mid.y, lo.y);
    };
    auto write_a = [&](unsigned dst, const uint2& hi, const uint2& mid, const uint2& lo) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 vh = {hi.x, hi.y}, vm = {mid.x, mid.y}, vl = {lo.x, lo.y};
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:32\n\tds_write_b64 %0, %3 offset:64"
                     :: "v"(dst), "v"(vh), "v"(vm), "v"(vl) : "memory");
    };
    // DMA instruction i (0 .. SUB * 25 - 1) of stage st: sub-chunk i / 25, KB i % 25 of its B image
    auto dma = [&](int st, int i) __attribute__((always_inline)) {
        const int sub = i / GG::B_DMAS, ins = i % GG::B_DMAS, ch = min(st * SUB + sub, nchunks - 1);
        __builtin_amdgcn_global_load_lds((glb_void*)(bsrc + (size_t)ch * bstep + ins * 1024),
                                         (lds_void*)(lds + (st & 1) * SSTAGE + sub * GG::STAGE + GG::A_BYTES + ins * 1024), 16, 0, 0);
    };
    constexpr int NDMA = (SUB * GG::B_DMAS + 7) / 8;             // DMA instructions per wave per stage (4 or 7)

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // One stage of one wave: SUB sub-chunks x NTL column tiles x 6 dependent MFMAs; the staging of the next stage goes
    // into the gaps between them (gap G counts MFMAs of the stage, skipping the last of each tile):
    //   G < NDMA: a B DMA instruction (wave w: w, w + 8, ..);  G NDMA ..: the float4s of A two stages ahead;
    //   then the split of the next stage's A and its ds_write_b64s.
    auto stage = [&](int st, auto ntl_t, float4 (&cur)[SUB], float4 (&nxt)[SUB]) __attribute__((always_inline)) {
        constexpr int NTL = decltype(ntl_t)::value;
        const bool has_next = st + 1 < nstages;
        const bool do_load = has_next && ABL != 2;
        const unsigned sb0 = lds_base + (st & 1) * SSTAGE;
        const unsigned adst = lds_base + ((st + 1) & 1) * SSTAGE + alds;
        uint2 hi[SUB], mid[SUB], lo[SUB];
        auto gap = [&](int G) __attribute__((always_inline)) {
            if (G < NDMA) {
                const int i = wu + 8 * G;
                if (do_load && i < SUB * GG::B_DMAS) dma(st + 1, i);
            } else if (G < NDMA + SUB) {
                if (do_load) load_a((st + 2) * SUB + (G - NDMA), nxt[G - NDMA]);
            } else if (G >= NDMA + SUB + 2 && G < NDMA + 2 * SUB + 2) {
                if (has_next) split4(cur[G - NDMA - SUB - 2], hi[G - NDMA - SUB - 2], mid[G - NDMA - SUB - 2], lo[G - NDMA - SUB - 2]);
            } else if (G >= NDMA + 2 * SUB + 2 && G < NDMA + 3 * SUB + 2) {
                const int q = G - NDMA - 2 * SUB - 2;
                if (has_next) write_a(adst + q * GG::STAGE, hi[q], mid[q], lo[q]);
            }
        };
#define DR_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            if (SUB > 1 && sub > 0 && st * SUB + sub >= nchunks) break;          // odd number of 16-chunks: uniform
            const unsigned sbase = sb0 + sub * GG::STAGE;
            const unsigned Ab = sbase + (wm * 32 + l31) * ROWB + 16 * h;
            const unsigned Bb = sbase + GG::A_BYTES + ((NTL == 4 ? 0 : 4) * 32 + l31) * ROWB + 16 * h;
            u32x4 a[3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) DR_LDS_READ(a[p], Ab, p * PL);
#pragma unroll
            for (int p = 0; p < 3; ++p) DR_LDS_READ(b[0][p], Bb, p * PL);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int cb = j & 1;
                if (j + 1 < NTL) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) DR_LDS_READ(b[cb ^ 1][p], Bb, (j + 1) * 32 * ROWB + p * PL);
                    asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[0]), a1 = __builtin_bit_cast(bf16x8, a[1]), a2 = __builtin_bit_cast(bf16x8, a[2]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[cb][0]), b1 = __builtin_bit_cast(bf16x8, b[cb][1]), b2 = __builtin_bit_cast(bf16x8, b[cb][2]);
#define DR_MFMA_GAP(X, Y, g)                                                              \
    if (ABL != 3) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Y, acc[j], 0, 0, 0); \
    else asm volatile("" ::"v"(X), "v"(Y));                                                \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (g < 5) { gap(5 * (sub * NTL + j) + g); __builtin_amdgcn_sched_barrier(0); }
                DR_MFMA_GAP(a2, b0, 0)
                DR_MFMA_GAP(a0, b2, 1)
                DR_MFMA_GAP(a1, b1, 2)
                DR_MFMA_GAP(a1, b0, 3)
                DR_MFMA_GAP(a0, b1, 4)
                DR_MFMA_GAP(a0, b0, 5)
#undef DR_MFMA_GAP
            }
        }
#undef DR_LDS_READ
    };

    // prologue: stage 0 staged, A of stage 1 in registers
    {
#pragma unroll
        for (int g = 0; g < NDMA; ++g) {
            const int i = wu + 8 * g;
            if (i < SUB * GG::B_DMAS) dma(0, i);
        }
#pragma unroll
        for (int q = 0; q < SUB; ++q) load_a(q, ra[0][q]);
#pragma unroll
        for (int q = 0; q < SUB; ++q) load_a(SUB + q, ra[1][q]);
#pragma unroll
        for (int q = 0; q < SUB; ++q) {
            uint2 hi, mid, lo;
            split4(ra[0][q], hi, mid, lo);
            write_a(lds_base + q * GG::STAGE + alds, hi, mid, lo);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int s0 = 0; s0 < nstages; s0 += 2) {
        if (wn == 0) stage(s0, std::integral_constant<int, 4>{}, ra[1], ra[0]);
        else stage(s0, std::integral_constant<int, 3>{}, ra[1], ra[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (s0 + 1 < nstages) {
            if (wn == 0) stage(s0 + 1, std::integral_constant<int, 4>{}, ra[0], ra[1]);
            else stage(s0 + 1, std::integral_constant<int, 3>{}, ra[0], ra[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // epilogue through LDS as in the 4-wave kernel; a wave owns a private [32][72] float region and moves two column
    // tiles per pass (rows of 256 bytes: a lane stores float4 (lane & 15) of row 4 it + (lane >> 4))
    {
        constexpr int EST = 72;
        float* const ep = reinterpret_cast<float*>(lds) + wu * 32 * EST;
        const int epi = P.epi, halfC = P.rot_C >> 1, rotC = P.rot_C, ldo = P.ldo;
        const float scale = P.scale;
        const float* __restrict__ bias = P.bias;
        const float* __restrict__ addend = P.addend;
        const float* __restrict__ cosT = P.cosT;
        const float* __restrict__ sinT = P.sinT;
        float* __restrict__ outp = P.out;
        const int ntl = wn ? 3 : 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (pass * 2 + jj >= ntl) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[((r & 3) + 8 * (r >> 2) + 4 * h) * EST + jj * 32 + l31] = acc[pass * 2 + jj][r];
            }
            const int c4 = (lane & 15) * 4;
            const int col = col0 + (wn * 4 + pass * 2) * 32 + c4;
            const bool col_ok = c4 < (ntl - pass * 2) * 32 && col < ncols;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias && col_ok) bv = *reinterpret_cast<const float4*>(bias + col);
            const int ridx = (epi & EPI_ROTARY) ? (col % rotC) >> 1 : 0;
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int rl = it * 4 + (lane >> 4), row = row0 + wm * 32 + rl;
                float4 v = *reinterpret_cast<const float4*>(ep + rl * EST + c4);
                if (row < rows && col_ok) {
                    if (epi & EPI_ROTARY) {
                        const float2 c = *reinterpret_cast<const float2*>(cosT + (size_t)row * halfC + ridx);
                        const float2 sn = *reinterpret_cast<const float2*>(sinT + (size_t)row * halfC + ridx);
                        const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
                        v.x = __fadd_rn(__fmul_rn(x0, c.x), __fmul_rn(-x1, sn.x));
                        v.y = __fadd_rn(__fmul_rn(x1, c.x), __fmul_rn(x0, sn.x));
                        v.z = __fadd_rn(__fmul_rn(x2, c.y), __fmul_rn(-x3, sn.y));
                        v.w = __fadd_rn(__fmul_rn(x3, c.y), __fmul_rn(x2, sn.y));
                    }
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (epi & EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
                    if (addend) {
                        const float4 ad = *reinterpret_cast<const float4*>(addend + (size_t)row * ldo + col);
                        v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                    }
                    *reinterpret_cast<float4*>(outp + (size_t)row * ldo + col) = v;
                }
            }
        }
    }
}

// W [ncols][K] fp32 -> the packed split image described above (one thread per 8 k of one row of one tile chunk)
__global__ void pack_weights_kernel(const float* __restrict__ W, char* __restrict__ out, int ncols, int K) {
    using GG = WideGeom;
    const int nck = (K + 15) / 16, tiles_n = (ncols + GG::BN - 1) / GG::BN;
    const size_t n = (size_t)nck * tiles_n * GG::BN * 2, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int half = (int)(i & 1);
    size_t rest = i >> 1;
    const int r = (int)(rest % GG::BN); rest /= GG::BN;
    const int tn = (int)(rest % tiles_n), ch = (int)(rest / tiles_n);
    const int col = tn * GG::BN + r, k = ch * 16 + half * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (col < ncols && k + e < K) ? W[(size_t)col * K + k + e] : 0.f;
    uint4 hi, mid, lo;
    split_pair(x[0], x[1], hi.x, mid.x, lo.x);
    split_pair(x[2], x[3], hi.y, mid.y, lo.y);
    split_pair(x[4], x[5], hi.z, mid.z, lo.z);
    split_pair(x[6], x[7], hi.w, mid.w, lo.w);
    char* d = out + ((size_t)ch * tiles_n + tn) * GG::B_IMG + (size_t)r * GG::ROWB + half * 16;
    *reinterpret_cast<uint4*>(d) = hi;
    *reinterpret_cast<uint4*>(d + GG::PL) = mid;
    *reinterpret_cast<uint4*>(d + 2 * GG::PL) = lo;
}

size_t gemm_packed_weight_bytes(int ncols, int K) {      // (the three-plane image is the larger one: room for either)
    return (size_t)((K + 15) / 16) * ((ncols + WideGeom::BN - 1) / WideGeom::BN) * WideGeom::B_IMG;
}

int launch_pack_weights(const float* W, int ncols, int K, void* out, hipStream_t st) {
    const size_t n = (size_t)((K + 15) / 16) * ((ncols + WideGeom::BN - 1) / WideGeom::BN) * WideGeom::BN * 2;
    if (n == 0) return DR_OK;
    // (the 16-byte row pads and the tail of each tile image are copied to LDS but never read as operands)
    if (gemm_f16x2()) {
        // image, then the ncols column factors 2^-s_c (the three-plane size that gemm_packed_weight_bytes reports has room)
        float* cinv = reinterpret_cast<float*>((char*)out + (size_t)((K + 15) / 16) * ((ncols + WideGeom2::BN - 1) / WideGeom2::BN) * WideGeom2::B_IMG);
        hipLaunchKernelGGL(wcol_scale_kernel, dim3((ncols + 3) / 4), dim3(256), 0, st, W, ncols, K, cinv);
        DR_LAUNCH_CHECK();
        hipLaunchKernelGGL(pack_weights_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, (char*)out, ncols, K, cinv);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, (char*)out, ncols, K);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

template <int ABL = 0>
static int configure_wide() {
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_wide_kernel<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)WideGeom::SMEM));
    return DR_OK;
}

template <int SUB, int ABL = 0>
static int configure_wide8() {
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_wide8_kernel<SUB, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(SUB * WideGeom::SMEM)));
    return DR_OK;
}

template <int SUB, int ABL = 0>
static int launch_wide8(const GemmBatch& g, hipStream_t st) {
    using GG = WideGeom;
    int maxt = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tl = ((g.p[i].rows + GG::BM - 1) / GG::BM + 7) / 8 * 8 * ((g.p[i].ncols + GG::BN - 1) / GG::BN);   // row blocks in groups of 8 (XCD map)
        maxt = tl > maxt ? tl : maxt;
    }
    if (maxt == 0) return DR_OK;
    double flops = 0;
    for (int i = 0; i < g.n; ++i) flops += 2.0 * g.p[i].rows * g.p[i].ncols * g.p[i].K;
    ProfScope ps(PK_GEMM_SPLIT, flops, st);
    hipLaunchKernelGGL((gemm_nt_wide8_kernel<SUB, ABL>), dim3(maxt, g.n), dim3(512), SUB * GG::SMEM, st, g);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

template <int ABL = 0>
static int launch_wide(const GemmBatch& g, hipStream_t st) {
    using GG = WideGeom;
    int maxt = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tl = ((g.p[i].rows + GG::BM - 1) / GG::BM + 7) / 8 * 8 * ((g.p[i].ncols + GG::BN - 1) / GG::BN);   // row blocks in groups of 8 (XCD map)
        maxt = tl > maxt ? tl : maxt;
    }
    if (maxt == 0) return DR_OK;
    double flops = 0;
    for (int i = 0; i < g.n; ++i) flops += 2.0 * g.p[i].rows * g.p[i].ncols * g.p[i].K;
    ProfScope ps(PK_GEMM_SPLIT, flops, st);
    hipLaunchKernelGGL((gemm_nt_wide_kernel<ABL>), dim3(maxt, g.n), dim3(GG::NT), GG::SMEM, st, g);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

static int launch_wide2(const GemmBatch& g, hipStream_t st) {
    using GG = WideGeom2;
    static bool attr_done = false;
    if (!attr_done) {
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_wide2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GG::SMEM));
        attr_done = true;
    }
    int maxt = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tl = ((g.p[i].rows + GG::BM - 1) / GG::BM + 7) / 8 * 8 * ((g.p[i].ncols + GG::BN - 1) / GG::BN);
        maxt = tl > maxt ? tl : maxt;
    }
    if (maxt == 0) return DR_OK;
    double flops = 0;
    for (int i = 0; i < g.n; ++i) flops += 2.0 * g.p[i].rows * g.p[i].ncols * g.p[i].K;
    ProfScope ps(PK_GEMM_SPLIT, flops, st);
    static const int no_amax = env_knob("DR_GEMM_NO_AMAX", 0);   // diagnostics: sweep always
    if (no_amax) {
        GemmBatch h = g;
        for (int i = 0; i < h.n; ++i) { if (no_amax & 1) { h.p[i].amax = nullptr; h.p[i].amax2 = nullptr; } if (no_amax & 2) h.p[i].omax = nullptr; }
        hipLaunchKernelGGL(gemm_nt_wide2_kernel, dim3(maxt, g.n), dim3(GG::NT), GG::SMEM, st, h);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    hipLaunchKernelGGL(gemm_nt_wide2_kernel, dim3(maxt, g.n), dim3(GG::NT), GG::SMEM, st, g);
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
    if (rc == DR_OK) rc = configure_wide<0>();
    if (rc == DR_OK) rc = configure_wide<1>();
    if (rc == DR_OK) rc = configure_wide<2>();
    if (rc == DR_OK) rc = configure_wide<3>();
    if (rc == DR_OK) rc = configure_wide<9>();
    if (rc == DR_OK) rc = configure_wide8<1>();
    if (rc == DR_OK) rc = configure_wide8<2>();
    if (rc == DR_OK) rc = configure_wide8<2, 2>();
    if (rc == DR_OK) rc = configure_wide8<2, 3>();
    return rc;
}

static int g_force_cfg = -1;   // tools / tests: force a configuration (0, 1, 2, 9: f32-MFMA tiles; 50..: wide split; 60..: 8-wave wide split)
void gemm_force_config(int c) { g_force_cfg = c; }

static int g_wide_min = -1;      // tests: force the threshold (-1 = environment / default)
void gemm_force_wide_min(int n) { g_wide_min = n; }
int gemm_wide_min_tiles() {
    static const int v = env_knob("DR_GEMM_WIDE_MIN", 128);
    return g_wide_min >= 0 ? g_wide_min : v;
}

// shapes / alignments the wide split kernel takes (everything else stays on the f32-MFMA kernels)
static bool wide_ok(const GemmProblem& p) {
    if (p.nbatch > 1) return false;
    if (!p.Wsplit || p.K % 8 || (p.A2 && p.K1 % 8) || p.ncols % 4 || p.ldo % 4) return false;
    if (((uintptr_t)p.out | (uintptr_t)p.addend | (uintptr_t)p.bias | (uintptr_t)p.Wsplit) & 15) return false;
    if ((p.epi & EPI_ROTARY) && (p.rot_C % 4 || ((uintptr_t)p.cosT | (uintptr_t)p.sinT) & 7)) return false;
    return true;
}

int launch_gemm(const GemmBatch& g, hipStream_t st) {
    if (g.n < 1 || g.n > 4) return DR_EINVAL;
    long nM = 0;                    // 64 x 64 tiles of the launch
    for (int i = 0; i < g.n; ++i) {
        const GemmProblem& p = g.p[i];
        if (p.K % 4 || p.lda % 4 || (p.A2 && (p.K1 % 4 || p.lda2 % 4))) return DR_ENOSUP;
        if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.A2) & 15) return DR_ENOSUP;
        nM += (long)((p.rows + 63) / 64) * ((p.ncols + 63) / 64) * (p.nbatch > 1 ? p.nbatch : 1);
    }
    // wide split-operand kernel: packed weights given and enough 128 x 224 tiles to occupy the chip
    long nW = 0;
    bool wide = true;
    for (int i = 0; i < g.n; ++i) {
        wide = wide && wide_ok(g.p[i]);
        nW += (long)((g.p[i].rows + 127) / 128) * ((g.p[i].ncols + 223) / 224);
    }
    const int wide_min = gemm_wide_min_tiles();
    // Launched alone, up to ~1.5 tiles per CU the 8-wave form (one workgroup per CU, two 16-chunks per barrier) is
    // 8-13 % faster (tools/gemm_split.py); inside the loop, where the engine keeps two batches in flight on two
    // streams, its 160 KB of LDS keeps the other stream's kernels off the CU and the 4-wave form wins by 2 %
    // (bench.py, DR_GEMM_WIDE8_MAX sweep) -- so it is opt-in.
    static const int wide8_max = env_knob("DR_GEMM_WIDE8_MAX", 0);
    // (weights packed in the two-plane fp16 mode can only be read by the two-plane kernel: whatever the tile count)
    // (and only deep reductions: below K = 128 the 2^-22 representation error is not hidden by the accumulation's own
    //  rounding -- those launches stay on the f32-input MFMA kernels, which read the fp32 weights)
    bool deep = true;
    for (int i = 0; i < g.n; ++i) deep = deep && g.p[i].K >= 128;
    if (wide && deep && gemm_f16x2() && g_force_cfg < 0 && nW >= wide_min) return launch_wide2(g, st);
    if (gemm_f16x2() && g_force_cfg < 0) wide = false;
    if (wide && nW >= wide_min && g_force_cfg < 0) return nW <= wide8_max ? launch_wide8<2>(g, st) : launch_wide<0>(g, st);
    // latency form: few tiles (the whole launch is one short wave of workgroups) and a k range that fits the registers
    int maxK = 0;
    long n32 = 0;
    for (int i = 0; i < g.n; ++i) {
        maxK = g.p[i].K > maxK ? g.p[i].K : maxK;
        n32 += (long)((g.p[i].rows + 31) / 32) * ((g.p[i].ncols + 31) / 32) * (g.p[i].nbatch > 1 ? g.p[i].nbatch : 1);
    }
    static const int direct_max = env_knob("DR_GEMM_DIRECT_MAX", 2048);   // (1193 rows x 1296 columns = 1558 tiles: 21 us against 36 us for the LDS-staged tiles)
    if (g_force_cfg < 0 && n32 <= direct_max && maxK <= 16 * 8 * 7) return maxK <= 8 * 8 * 7 ? launch_direct<8, 7>(g, st) : launch_direct<16, 7>(g, st);
    int cfg = nM >= 128 ? 9 : 0;     // 9 = 64 x 64 tiles with a single LDS buffer (18 KB -> 8 workgroups per CU): best of
                                     // every f32-MFMA configuration measured on the loop's shapes (tools/gemm_bench.py)
    static const int env_cfg = env_knob("DR_GEMM_CFG", -1);   // tools/: tile experiments
    if (env_cfg >= 0 && cfg == 9) cfg = env_cfg;
    if (g_force_cfg >= 0) cfg = g_force_cfg;
    for (int i = 0; i < g.n; ++i)
        if (g.p[i].nbatch > 1 && cfg >= 20) return DR_ENOSUP;   // strided batches: f32-MFMA kernels only
    if (cfg >= 50 && cfg < 80 && (cfg >= 70 || gemm_f16x2())) {   // 70: the two-plane fp16 kernel; in that packing mode
        for (int i = 0; i < g.n; ++i)                              // every packed configuration means it (the image is its)
            if (!wide_ok(g.p[i])) return DR_ENOSUP;
        if (!gemm_f16x2()) return DR_EINVAL;                       // (weights packed in the three-plane mode)
        return launch_wide2(g, st);
    }
    if (cfg >= 50 && cfg < 60) {
        for (int i = 0; i < g.n; ++i)
            if (!wide_ok(g.p[i])) return DR_ENOSUP;
        if (cfg == 50) return launch_wide<0>(g, st);
        if (cfg == 51) return launch_wide<1>(g, st);
        if (cfg == 52) return launch_wide<2>(g, st);
        if (cfg == 53) return launch_wide<3>(g, st);
        if (cfg == 59) return launch_wide<9>(g, st);
    }
    if (cfg >= 60 && cfg < 70) {
        for (int i = 0; i < g.n; ++i)
            if (!wide_ok(g.p[i])) return DR_ENOSUP;
        if (cfg == 60) return launch_wide8<1>(g, st);
        if (cfg == 61) return launch_wide8<2>(g, st);
        if (cfg == 62) return launch_wide8<2, 2>(g, st);
        if (cfg == 63) return launch_wide8<2, 3>(g, st);
    }
    if (cfg == 11) return launch_direct<8, 7>(g, st);
    if (cfg == 12) return launch_direct<16, 7>(g, st);
    if (cfg == 9) return launch_cfg<CFG_M1B>(g, st);
    if (cfg == 2) return launch_cfg<CFG_LARGE>(g, st);
    if (cfg == 1) return launch_cfg<CFG_MEDIUM>(g, st);
    return launch_cfg<CFG_SMALL>(g, st);
}

}  // namespace dr
