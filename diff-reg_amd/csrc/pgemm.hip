// pgemm.hip -- the layer nn.Linears of the denoising loop (q/k/v/merge/mlp of 3D/models/transformero.py:26-96, src_proj of
// 3D/models/matching.py:107) as a GEMM whose operands are BOTH pre-split fp16 hi / lo plane images in global memory.
//
// out = x W^T in fp32 accuracy from three fp16 MFMA products per fp32 MAC (hi*hi + hi*lo + lo*hi, fp32 accumulate; see
// gemm.hip for the numerics of the two-plane split).  What changes against gemm_nt_wide2_kernel is where the planes come from:
// the PRODUCER of an activation (this kernel's own epilogue, the attention kernel, planes_from_f32 for the external
// features) writes them, scaled by an exact power of two from a per-row upper bound that is propagated analytically
// (|x W^T| <= bound(x) max_c ||W_c||_1, |LayerNorm| <= sqrt(C) max|gamma| + max|beta|, |softmax V| <= max|V|), so that no
// kernel ever sweeps a row for its maximum and the A operand streams by LDS-DMA exactly like the weights: no VGPR-held
// requests, no VALU split, no in-order vmcnt chain between A loads and the B image.
//
// Two main loops share the epilogue.  Round 4 (M16; the 128-row geometry of <= 448 columns): v_mfma_f32_16x16x32_f16 on chunk PAIRS, 8 waves as
// 2 (rows) x 4 (columns) of 64 x 112, the weight fragment as the MFMA's first operand so that the accumulators are in the epilogue's layout --
// see the comment at `run16`.  Rounds 2-3 (the 64-row workgroups, the 256- and 576-column geometries): v_mfma_f32_32x32x16_f16, described next.
//
// Geometry: a workgroup is 128 rows x one column block of up to 448 columns (the whole C = 432 row of one nn.Linear), 8 waves
// as 4 (rows) x 2 (columns), a wave = 32 x 224 = 7 accumulator tiles of v_mfma_f32_32x32x16_f16; one workgroup per CU.
// A stage = one 16-deep k-chunk = 8 KB of A + 28 KB of W, 36 one-KB DMA instructions dealt over the 8 waves; NST = 4 stages
// ring in LDS (144 KB), the DMA of stage s + 3 is issued in the middle of stage s, ONE barrier per stage placed mid-stage:
//   tiles 0..2 of stage s | wait: own DMAs of stage s + 1 landed | s_barrier | issue DMAs of stage s + 3 into the slot of
//   stage s - 1 (every wave is past it) | tiles 3..6 | fragments of stage s + 1 are read during the last tile
// so no MFMA ever waits behind a barrier for a fragment read, and a DMA has two stage times to land.
// Full rows per workgroup make the epilogue the place where LayerNorm (+ residual) happens (merge -> norm1, mlp2 -> norm2 + x):
// the accumulators are transposed through LDS so that a lane owns 56 columns of one row, the row statistics cross the two
// column waves through LDS, and the result leaves as fp32 rows and / or as the plane image of the next GEMM.
#include <cstdlib>
#include <string.h>
#include <type_traits>
#include "pgemm.h"

namespace dr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// x0, x1 -> packed (hi0, hi1), (lo0, lo1): hi = fp16(x) round to nearest even, lo = fp16(x - hi)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 f = {x0, x1};
    const f16x2 h = __builtin_convertvector(f, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 r = {x0 - hf.x, x1 - hf.y};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// exponent s with 2^14 <= m 2^s < 2^15 (0 for m = 0 / inf / nan), clamped so that 2^s and 2^-s are normal floats
__device__ __forceinline__ int scale_exp(float m) {
    const unsigned bits = __float_as_uint(m);
    const int e = (int)((bits >> 23) & 0xff) - 127;
    const bool ok = m > 0.f && e < 128;
    return ok ? min(max(14 - e, -100), 100) : 0;
}
__device__ __forceinline__ float pow2i(int s) { return __uint_as_float((unsigned)(127 + s) << 23); }
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
}
__device__ __forceinline__ float dpp_xor1f(float v) { return __uint_as_float(dpp_xor1(__float_as_uint(v))); }
__device__ __forceinline__ float dpp_xor2f(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xF, 0xF, true));   // [2, 3, 0, 1]
}
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// WMN = waves along the rows: 4 -> 128-row workgroups of 512 threads; 2 -> 64-row workgroups of 256 threads (same wave tile, same ring
// depth): twice the workgroups for launches that would leave CUs idle (fewer 128-row workgroups than CUs: small batches, BASELINE cfg3's 8 pairs)
template <int TNW_, int NST_, int WMN_ = 4>
struct PgGeom {
    static constexpr int TNW = TNW_, NST = NST_, WMN = WMN_;
    static constexpr int BM = 32 * WMN, BNW = 32 * TNW, BN = 2 * BNW, NTHR = 128 * WMN;
    static constexpr int A_ST = BM * 64, B_ST = BN * 64, STAGE = A_ST + B_ST;
    static constexpr int A_IMG = 128 * 64;                  // bytes between two k-chunks of a 128-row block in an A image
    static constexpr int NA = A_ST / 1024;                  // DMA instructions of the A part of a stage (8 / 4)
    static constexpr int NB = B_ST / 1024;                  // DMA instructions of the weight part of a stage (28)
    static constexpr int NFULL = NB / 8, REM = NB % 8;      // every wave issues 1 (A) + NFULL, waves < REM one more
    static constexpr int RING = NST * STAGE;
    static constexpr int NI = BNW / 16;                     // float4 pieces of a row a lane holds after the transposition (14)
    static constexpr int EP_S = (BNW - 48 + 63) / 64 * 64 + 48;   // row stride of the transposition, = 48 mod 64 floats: conflict-free float4 reads
    static constexpr int EP_BYTES = 2 * WMN * 16 * EP_S * 4;
    static constexpr int WORK = RING > EP_BYTES ? RING : EP_BYTES;
    // + fac[128], rinv[128], partial sums [2][128], partial squares [2][128], gamma | beta | bias | cinv [4][BN], the rows' image bounds [128]
    static constexpr int SMEM = WORK + 7 * 128 * 4 + 4 * BN * 4;
    static constexpr int SMEM16 = WORK + 11 * 128 * 4 + 4 * BN * 4; // the 16x16x32 form: four column waves -> partial sums / squares [4][128]
    static constexpr int MID = (TNW - 1) / 2 - 1 < 0 ? 0 : (TNW - 1) / 2 - 1;   // the barrier sits after this tile (2 of 7)
};

#define PG_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define PG_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n) : "memory")

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TNW, int NST, int MODE, int WMN = 4, bool M16 = false>
__global__ __launch_bounds__(128 * WMN) __attribute__((amdgpu_waves_per_eu(WMN / 2, WMN / 2))) void pgemm_kernel(PgBatch G) {
    using GG = PgGeom<TNW, NST, WMN>;
    constexpr int BNW = GG::BNW, BN = GG::BN, STAGE = GG::STAGE, A_ST = GG::A_ST, NI = GG::NI, EP_S = GG::EP_S, BM = GG::BM, NTHR = GG::NTHR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;

    const PgProblem& P = G.p[blockIdx.y];
    // KSPLIT (64-row LayerNorm launches that would leave half the chip idle; launch_pgemm sets P.ksplit): the k range of a row block is split
    // over TWO workgroups -- dealt like two column blocks, ids 8 apart: the same XCD -- which exchange half of their partial sums behind the
    // main loop and each finish 16 of a wave's 32 rows (see the exchange in front of the epilogue)
    constexpr bool KS_OK = !M16 && WMN == 2 && MODE == PG_LN;
    const bool ksplit = KS_OK && P.ksplit != 0;
    const int rows = P.rows, C = P.C, nblk = ksplit ? 2 : P.nblk, nc1 = P.A1 ? P.nc1 : 0;
    const int rbs = (rows + BM - 1) / BM;                       // BM-row blocks; block rb = rows rb BM .. of the 128-row image block rb BM / 128
    // workgroup id -> (row block, column block), XCD-aware: ids are dealt round-robin to the 8 XCDs, so the column blocks of a
    // row block get ids 8 apart: they share an L2 and the A rows cross the fabric once
    const int grp = blockIdx.x / (8 * nblk), rem = blockIdx.x % (8 * nblk);
    const int rb = grp * 8 + (rem & 7), khalf = ksplit ? (rem >> 3) : 0, nb = ksplit ? 0 : (rem >> 3);
    if (rb >= rbs) return;
    // this workgroup's k-chunks: all of them, or (KSPLIT: one segment only) the first / second half
    const int kbeg = khalf ? (P.nc0 + 1) / 2 : 0;
    const int nc0 = ksplit ? (khalf ? P.nc0 - kbeg : (P.nc0 + 1) / 2) : P.nc0, nst = nc0 + nc1;
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w % WMN, wn = w / WMN;
    const int rb128 = (rb * BM) >> 7, sub = (rb * BM) & 127;      // image block and first row inside it (0 / 64)

    float* const s_fac = reinterpret_cast<float*>(lds + GG::WORK);   // 2^(s1 - s0) of the rows (two-segment A operand)
    float* const s_rinv = s_fac + 128;                               // 2^-s of the rows (last segment)
    constexpr int NCW = M16 ? 4 : 2;                                 // column waves of a row (LayerNorm partials)
    float* const s_sum = s_rinv + 128;                               // [NCW][128] LayerNorm partial sums of the column waves
    float* const s_sq = s_sum + NCW * 128;                           // [NCW][128] partial squares
    // nn.Linear bias of the block: staged in LDS (behind gamma | beta) and read piece by piece in the epilogue -- held in registers
    // (14 float4 per lane) it made the 128-row q|k|v / mlp0 instantiations spill
    const bool has_bias = P.bias != nullptr;
    float* const s_bias = s_sq + NCW * 128 + 2 * BN;
    float* const s_gam = s_sq + NCW * 128;                           // gamma | beta of the block (PG_LN)
    float* const s_bet = s_gam + BN;
    float* const s_cinv = s_bias + BN;                               // 2^-s_c of the block's columns
    float* const s_bound = s_cinv + BN;                              // [128] bound of the row's output image
    const bool rot = MODE != PG_LN && ((P.rot_mask >> nb) & 1);
    const bool per_blk = P.pimg_blk_stride != 0;
    // (the 16x16x32 loop calls this BEHIND the DMA of its first chunk pair: the loads' latency then passes while the pair is in flight)
    auto stage_inputs = [&]() __attribute__((always_inline)) {
        if (has_bias) {
            const float* bp = P.bias + (size_t)nb * C;
            for (int c = t; c < BN; c += NTHR) s_bias[c] = c < C ? bp[c] : 0.f;
        }
        // Everything else the epilogue needs from memory that does not depend on the accumulators is fetched HERE, in front of the main loop:
        // the columns' 2^-s_c, gamma | beta, and the bound each row's image is scaled by.  (Round 4: the bound's loads -- row bounds, weight norms,
        // group bounds: a chain of dependent loads per round of rows, each behind the previous round's stores -- were ~10 exposed memory latencies
        // per round in the epilogue.)
        {
            const float* cp = P.W.cinv + (size_t)nb * BN;
            for (int c = t; c < BN; c += NTHR) s_cinv[c] = cp[c];
            if (MODE == PG_LN && P.gamma)
                for (int c = t; c < BN; c += NTHR) {
                    s_gam[c] = c < C ? P.gamma[c] : 0.f;
                    s_bet[c] = c < C ? P.beta[c] : 0.f;
                }
        }
        // the bound of the row's GROUP (blocks flagged in grp_mask) without a kernel of its own: when a group's rows are a whole number of
        // workgroups (grp_rows % 128 == 0: the caller then passes grp_bnd = nullptr) the workgroup takes the maximum over its group's rows here
        float gmax = 0.f;
        if (MODE != PG_LN && ((P.grp_mask >> nb) & 1) && !P.grp_bnd) {
            const int gbase = (rb * BM) / P.grp_rows * P.grp_rows;
            float m = 0.f;
            for (int i = t; i < P.grp_rows; i += NTHR) m = fmaxf(m, P.bnd0[gbase + i]);
            m = wave_max(m);
            if (lane == 0) s_sum[w] = m;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NTHR / 64; ++k) gmax = fmaxf(gmax, s_sum[k]);
        }
        if (t < BM) {
            const int row = min(rb * BM + t, rows - 1);
            const float b0 = P.bnd0[row], b1 = nc1 > 0 ? P.bnd1[row] : 0.f;
            const int e0 = scale_exp(b0);
            int e1 = e0;
            if (nc1 > 0) e1 = scale_exp(b1);
            s_fac[t] = pow2i(min(max(e1 - e0, -120), 120));
            s_rinv[t] = pow2i(-e1);
            if (P.pimg) {
                float bound;
                if (MODE == PG_LN) bound = ((P.bnd_res && !P.ln_postadd) ? P.bnd_res[row] : 0.f) + (P.lnB ? P.lnB[0] : 0.f);
                else {
                    // |x W^T| <= bound(x) max_c ||W_c||_1 (x sqrt 2 behind the rotary embedding, x |scale|); blocks flagged in grp_mask take
                    // the bound of the row's GROUP (a pair's side) so that all rows of a group share one scale (the attention kernel's K / V)
                    const float bin = ((P.grp_mask >> nb) & 1) ? (P.grp_bnd ? P.grp_bnd[P.grp_first + row / P.grp_rows] : gmax) : fmaxf(b0, b1);
                    // blocks that share ONE image and ONE bound array (mlp0's two column blocks -> hid) must derive the same scale: the bound is
                    // taken from the largest of their weight norms (a per-block bound would scale block 1 by 2^s(bin wnorm[1]) while every
                    // consumer rescales the whole row by the stored 2^-s(bin wnorm[0]): off by a power of two where the two straddle one)
                    float wn_ = P.W.wnorm[nb];
                    if (!per_blk)
                        for (int b2 = 0; b2 < nblk; ++b2) wn_ = fmaxf(wn_, P.W.wnorm[b2]);
                    float bm = 0.f;                                  // |x W^T + b| <= bound(x) ||W|| + max |b|  (same rule for blocks sharing an image)
                    if (P.bias_max) {
                        bm = P.bias_max[nb];
                        if (!per_blk)
                            for (int b2 = 0; b2 < nblk; ++b2) bm = fmaxf(bm, P.bias_max[b2]);
                    }
                    bound = (bin * wn_ + bm) * (rot ? 1.41421366f : 1.f) * fabsf(P.scale);
                }
                s_bound[t] = bound;
            }
        }
    };

    // ---- fragment addresses: a lane reads 16 bytes = 8 k of "its" row; lane half h takes k 8 h .. 8 h + 7
    const int sw = (l31 >> 2) & 3;
    const unsigned offAh = (wm * 32 + l31) * 64 + ((h ^ sw) << 4), offAl = (wm * 32 + l31) * 64 + (((2 + h) ^ sw) << 4);
    const unsigned offBh = A_ST + (wn * BNW + l31) * 64 + ((h ^ sw) << 4), offBl = A_ST + (wn * BNW + l31) * 64 + (((2 + h) ^ sw) << 4);

    f32x16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    u32x4 fa0[2], fa1[2], fx[TNW / 2][2], fy[TNW - TNW / 2][2];
    f32x4 c16[M16 ? 4 : 1][M16 ? 7 : 1];                        // M16: the wave's 4 x 7 accumulator tiles of 16 x 16

    // The two waves of a SIMD (w and w + 4: the column halves wn = 0 / 1 of one 32-row strip) run the same MFMA stream -- burst 1 = tiles
    // 0 .. TNW / 2 - 1, the stage's one barrier, burst 2 = the other tiles -- and differ in what they feed to the DMA:
    //   128-row workgroups on a 4-slot ring (ALL0): group 0 (wn = 0) issues every piece of stage s + 2 in the gaps of burst 1 of stage s,
    //     group 1 only computes;
    //   64-row workgroups and 3-slot rings: group 0 issues its half of the weight pieces in burst 1 (4 slots: stage s + 2) or burst 2
    //     (3 slots), group 1 the A block and the other half in burst 2 (stage s + NST - 1).
    // (In lockstep -- both waves of a SIMD issuing in the same gaps -- the halves serialise: 2.2 us per stage pair against 1.04 us of
    // staging alone and 1.2 us of MFMAs alone.  What the staging costs is the bytes it moves into the LDS, r03_pgemm_overlap_ablation.json.)
    auto run = [&](auto grp_t) __attribute__((always_inline)) {
        constexpr int GRP = decltype(grp_t)::value;
        // EARLY: group 0 issues stage s + NST - 2 in the FIRST burst of stage s (its slot is free since the barrier of stage s - 1).
        // That needs a slot beyond the one landing for stage s + 1, i.e. NST >= 4; with a 3-slot ring (the 576-column geometry)
        // both groups issue stage s + 2 in the second burst and wait for everything at the next barrier.
        constexpr bool EARLY = GRP == 0 && NST >= 4;
        constexpr int AHEAD = EARLY ? NST - 2 : NST - 1;             // stages in flight beyond the current one after the prologue
        // ALL0 (128-row workgroups on a 4-slot ring, round 3): group 0 issues EVERY piece of a stage -- its 2 A pieces and 7 weight pieces
        // per wave, one per gap of the first burst -- and group 1 only computes.  Measured on one box against the even split below (A +
        // 10 weight instructions from group 1 in the second burst, 18 from group 0 in the first): qkv 357 -> 348 us, merge 95.3 -> 92.2,
        // mlp0 280.7 -> 271.1, mlp2 190.9 -> 185.5 per launch at 65 536 rows (shifting 6 or 10 weight instructions to group 0 lands in
        // between; group 1 issuing its even share before the barrier too, in alternate gaps, is 3 % SLOWER than the even split's
        // successor here): one issuing wave per SIMD pair whose partner never issues is what pays, not the burst the pieces sit in.
        constexpr bool ALL0 = NST >= 4 && WMN == 4;
        constexpr int AG = ALL0 ? 0 : 1;                             // the group that copies the A block
        constexpr int WB_CNT = ALL0 ? 0 : (GG::NB + GG::NA) / 2 - GG::NA;       // weight instructions of group 1
        constexpr int NW0 = GG::NB - WB_CNT;                         // weight instructions of group 0
        constexpr int NAP = GRP == AG ? 2 : 0;                       // A pieces of a wave of this group
        constexpr int NPIECE = GRP ? NAP + (WB_CNT + WMN - 1) / WMN : NAP + (NW0 + WMN - 1) / WMN;
        constexpr int NFULLP = GRP ? NAP + WB_CNT / WMN : NAP + NW0 / WMN;   // pieces every wave of the group issues
        constexpr int PSTEP = (2 * NPIECE <= 3 * (TNW - TNW / 2)) ? 2 : 1;   // gaps between two pieces issued in the second burst
        static_assert(EARLY || PSTEP * NPIECE <= 3 * (TNW - TNW / 2), "the DMA pieces of a stage must fit the gaps of the second burst");
        static_assert(!EARLY || NPIECE <= 3 * (TNW / 2), "... and those of the early group the gaps of the first");
        constexpr int REMP = GRP ? WB_CNT % WMN : NW0 % WMN;         // waves (local index) < REMP issue one more
        const int wl = w % WMN;
        // DMA pieces of this wave (1 KB instructions; `ins` = index inside the A block / the weight block of a stage):
        //   the A group (group 0 when ALL0, else group 1): A instructions 2 wl, 2 wl + 1, then its weight instructions st0 ..
        //   even split (64-row workgroups, 3-slot rings): group 1 A + WB_CNT weight instructions, group 0 the other NB - WB_CNT
        // issued in the SGPR-base + VGPR-offset + immediate form: per piece an s_mov to M0 and the load, nothing else
        const int st0 = GRP ? wl * (WB_CNT / WMN) + min(wl, WB_CNT % WMN) : WB_CNT + wl * (NW0 / WMN) + min(wl, NW0 % WMN);
        const unsigned voffW = lane * 16 + st0 * 1024, voffA = lane * 16 + 2 * wl * 1024;
        const char* ga = P.A0 + ((size_t)rb128 * P.nc0 + kbeg) * GG::A_IMG + sub * 64;   // A block of stage ti (wave-uniform)
        const char* gb = P.W.img + ((size_t)nb * (P.nc0 + nc1) + kbeg) * GG::B_ST;       // weight block of stage ti
        const char* const ga1 = nc1 > 0 ? P.A1 + (size_t)rb128 * nc1 * GG::A_IMG + sub * 64 : nullptr;
        int ti = 0;                                                  // next stage this wave issues
        // (the instruction's immediate offset is added to the global address AND to the LDS address M0 + 16 lane)
#define PG_DMA(ldsaddr, voff, gbase, imm)                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(ldsaddr), "v"(voff), "s"(gbase), "i"(imm) : "memory")
        auto dma_piece = [&](int p, unsigned dstb) __attribute__((always_inline)) {   // dstb = LDS address of the slot of stage ti
            const unsigned m0w = dstb + A_ST + st0 * 1024;
            // weight piece pw of a wave: the immediate reaches 3 KB, every 4 pieces move the register base by 4 KB
            const int pw = p - NAP;
            const unsigned hop = (unsigned)(pw >> 2) * 4096;
            if (p < NAP) PG_DMA(dstb + 2 * wl * 1024, voffA, ga, p * 1024);
            else if (pw < (GRP ? WB_CNT : NW0) / WMN || wl < (GRP ? WB_CNT : NW0) % WMN) PG_DMA(m0w + hop, voffW + hop, gb, (pw & 3) * 1024);
        };
        auto dma_advance = [&]() __attribute__((always_inline)) {
            ++ti;
            gb += GG::B_ST;
            ga = (ti == nc0) ? ga1 : ga + GG::A_IMG;
        };

        // One stage = two MFMA bursts with no wait inside: tiles 0 .. NT1 - 1 (fragment set X), then tiles NT1 .. TNW - 1 (set Y).
        // The fragment reads of a burst are issued in the MFMA gaps of the burst BEFORE it (Y during burst 1; the next stage's
        // A fragments and X during burst 2), so a read has a whole burst (~300 cycles alone, twice that beside the partner
        // wave) to return and a wave parks only at the stage's barrier.  (With reads one tile ahead both waves of a SIMD sat in
        // s_waitcnt lgkmcnt at the same time: 34 % of the wave cycles parked, the MFMA pipe 48 % busy.)
        auto stage = [&](int s, auto steady_t, u32x4 (&ac)[2], u32x4 (&an)[2]) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady_t)::value;   // every condition of the tail is known true
            constexpr int NT1 = TNW / 2, NT2 = TNW - NT1, M1 = 3 * NT1;      // M1 = MFMAs of burst 1
            const unsigned sb = lds_base + (unsigned)(s % NST) * STAGE, sbn = lds_base + (unsigned)((s + 1) % NST) * STAGE;
            const unsigned Bh = sb + offBh, Bl = sb + offBl, Bhn = sbn + offBh, Bln = sbn + offBl;
            const bool has_next = STEADY || s + 1 < nst;
            const bool do_issue = STEADY || ti < nst;
            const unsigned dstb = lds_base + (unsigned)(ti % NST) * STAGE;
            if (s == nc0 && nc1 > 0) {
                // second A segment starts: bring the accumulators from the scale of segment 0 to that of segment 1 (exact powers of two)
                const unsigned fb0 = lds_base + GG::WORK + (wm * 32 + 4 * h) * 4;
                u32x4 f[4];
                PG_READ(f[0], fb0, 0); PG_READ(f[1], fb0, 32); PG_READ(f[2], fb0, 64); PG_READ(f[3], fb0, 96);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TNW; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] *= __uint_as_float(f[r >> 2][r & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            auto gap = [&](int m) __attribute__((always_inline)) {
                // ---- fragment reads of the next burst: two per gap, in the first gaps of the burst (the wait that ends the burst
                // then finds them landed: the last read is >= 4 MFMAs old)
                {
                    if (m < NT2) {                                       // set Y: tile NT1 + m
                        PG_READ(fy[m][0], Bh, (NT1 + m) * 2048);
                        PG_READ(fy[m][1], Bl, (NT1 + m) * 2048);
                    } else if (m == M1 && has_next) {                    // next stage's A fragments
                        PG_READ(an[0], sbn + offAh, 0);
                        PG_READ(an[1], sbn + offAl, 0);
                    } else if (m > M1 && m <= M1 + NT1 && has_next) {    // next stage's set X
                        PG_READ(fx[m - M1 - 1][0], Bhn, (m - M1 - 1) * 2048);
                        PG_READ(fx[m - M1 - 1][1], Bln, (m - M1 - 1) * 2048);
                    }
                }
                // ---- DMA pieces: group 0 in burst 1 (stage s + NST - 2), group 1 in burst 2 (stage s + NST - 1)
                constexpr int MF = EARLY ? 0 : (PSTEP == 2 ? M1 + 1 : M1), STEP = EARLY ? 1 : PSTEP;
                if (m >= MF && m < MF + STEP * NPIECE && (m - MF) % STEP == 0) {
                    if (do_issue) dma_piece((m - MF) / STEP, dstb);
                    if ((m - MF) / STEP == NPIECE - 1 && do_issue) dma_advance();
                }
                // ---- the stage's barrier, between the bursts
                if (m == M1 - 1) {
                    if (has_next) {
                        // own DMAs of stage s + 1 have landed (those of stage s + 2, issued later, may still fly)
                        if (NST >= 4 && (STEADY || s + 2 < nst)) {
                            if (wl < REMP) PG_VMCNT(NFULLP + 1); else PG_VMCNT(NFULLP);
                        } else {
                            PG_VMCNT(0);
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // set Y has arrived (issued during burst 1)
                }
            };
            const f16x8 ah = __builtin_bit_cast(f16x8, ac[0]), al = __builtin_bit_cast(f16x8, ac[1]);
#define PG_MFMA(X, Y, g)                                                        \
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X, Y, acc[j], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                          \
    gap(3 * j + g);                                                             \
    __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // A fragments and set X of this stage (issued during the previous burst 2)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT1; ++j) {
                const f16x8 bh = __builtin_bit_cast(f16x8, fx[j][0]), bl = __builtin_bit_cast(f16x8, fx[j][1]);
                PG_MFMA(al, bh, 0)                               // smallest terms first
                PG_MFMA(ah, bl, 1)
                PG_MFMA(ah, bh, 2)
            }
#pragma unroll
            for (int j = NT1; j < TNW; ++j) {
                const f16x8 bh = __builtin_bit_cast(f16x8, fy[j - NT1][0]), bl = __builtin_bit_cast(f16x8, fy[j - NT1][1]);
                PG_MFMA(al, bh, 0)
                PG_MFMA(ah, bl, 1)
                PG_MFMA(ah, bh, 2)
            }
#undef PG_MFMA
        };

        // ---- prologue: group 1 puts stages 0 .. NST - 2 in flight, group 0 stages 0 .. NST - 3 (it issues stage s + NST - 2 in
        // the first burst of stage s, group 1 stage s + NST - 1 in the second); stage 0 landed
        // (the epilogue's inputs are fetched behind stage 0's pieces: their latency passes while the stage is in flight, and the wait for
        // them is the wait for stage 0)
#pragma unroll
        for (int q2 = 0; q2 < AHEAD; ++q2) {
            const unsigned dstb = lds_base + (unsigned)q2 * STAGE;
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) dma_piece(i, dstb);
            dma_advance();
            if (q2 == 0) {
                __builtin_amdgcn_sched_barrier(0);
                stage_inputs();
                __builtin_amdgcn_sched_barrier(0);
                PG_VMCNT(0);
            }
        }
        __builtin_amdgcn_s_barrier();
        PG_READ(fa0[0], lds_base + offAh, 0);
        PG_READ(fa0[1], lds_base + offAl, 0);
#pragma unroll
        for (int j = 0; j < TNW / 2; ++j) {
            PG_READ(fx[j][0], lds_base + offBh, j * 2048);
            PG_READ(fx[j][1], lds_base + offBl, j * 2048);
        }
        int s = 0;
        for (; s + 1 < nst - (NST - 1); s += 2) {                    // steady state: every stage issues, has a successor
            stage(s, std::true_type{}, fa0, fa1);
            stage(s + 1, std::true_type{}, fa1, fa0);
        }
        for (; s < nst; s += 2) {                                    // tail
            stage(s, std::false_type{}, fa0, fa1);
            if (s + 1 < nst) stage(s + 1, std::false_type{}, fa1, fa0);
        }
    };
    // ---- M16: the same product on v_mfma_f32_16x16x32_f16 (round 4; the guide's DVFS notes: the chip holds a higher clock on this shape, and
    // fewer LDS read bytes per MFMA raise it further).  A wave = 64 rows x 112 columns = 4 x 7 tiles of 16 x 16 (22 fragment reads per 84 MFMAs
    // instead of 32); one MFMA contracts TWO k-chunks: lane group g = lane / 16 reads unit (g >> 1) of chunk c + (g & 1) -- with this k order
    // the 16-row b128 fragment read is conflict-free on the image's unit swizzle (the read's lane groups {0-3, 12-15, 20-27}, .. hit 16
    // different (row % 4, position) pairs; with the natural order k = 8 g .. it is 2-way).  A stage = a chunk PAIR (72 KB): two buffers in
    // the 4-slot ring, ONE barrier per pair, in front of the last of the stage's four passes (pass i = row tile i x all 7 column tiles x 3
    // products); behind it the pair's slots are free and the next pair has landed: the last pass reads the next stage's weight fragments
    // into the registers its own MFMAs have just released, the issuing waves (one per SIMD, as ALL0 above) feed the DMA of pair p + 2
    // in that pass and in the next stage's first two.
    // A segment with an odd number of chunks is closed by a VIRTUAL chunk: nothing is copied, its lane groups' A fragments are zeroed
    // (the slot's stale weight planes are finite), so pairs stay aligned with the ring and with the segment boundary's rescale.
    auto run16 = [&](auto iss_t) __attribute__((always_inline)) {
        constexpr bool ISS = decltype(iss_t)::value;
        const int l15 = lane & 15, g = lane >> 4, cs = g & 1, uh = g >> 1, sw16 = (l15 >> 2) & 3;
        const int wn4 = w & 3, wm2 = w >> 2;
        const unsigned oAh = cs * STAGE + (wm2 * 64 + l15) * 64 + ((uh ^ sw16) << 4), oAl = cs * STAGE + (wm2 * 64 + l15) * 64 + (((2 + uh) ^ sw16) << 4);
        const unsigned oBh = cs * STAGE + A_ST + (wn4 * 112 + l15) * 64 + ((uh ^ sw16) << 4), oBl = cs * STAGE + A_ST + (wn4 * 112 + l15) * 64 + (((2 + uh) ^ sw16) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) c16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 fb[7][2], fa[2][2];
        const int nv0 = (nc0 + 1) & ~1, nvt = nv0 + ((nc1 + 1) & ~1), npair = nvt >> 1;      // virtual chunk counts (segments padded to pairs)
        const int z0 = (nc0 & 1) ? nv0 - 1 : -1, z1 = (nc1 & 1) ? nvt - 1 : -1;              // the virtual (zero) chunks
        const int wl = w & 3, st0 = wl * 7;
        const unsigned voffW = lane * 16 + st0 * 1024, voffA = lane * 16 + 2 * wl * 1024;
        const char* ga = P.A0 + (size_t)rb128 * nc0 * GG::A_IMG + sub * 64;
        const char* gb = P.W.img + (size_t)nb * nst * GG::B_ST;
        const char* const ga1 = nc1 > 0 ? P.A1 + (size_t)rb128 * nc1 * GG::A_IMG + sub * 64 : nullptr;
        int ti = 0, tr = 0;                                              // next virtual chunk to issue, real chunks issued
        auto piece = [&](int q) __attribute__((always_inline)) {        // q = 0 .. 8 of chunk ti
            if (ti == z0 || ti == z1) return;
            const unsigned dstb = lds_base + (unsigned)(ti % NST) * STAGE;
            const unsigned m0w = dstb + A_ST + st0 * 1024;
            const int pw = q - 2;
            const unsigned hop = (unsigned)(pw >> 2) * 4096;
            if (q < 2) PG_DMA(dstb + 2 * wl * 1024, voffA, ga, q * 1024);
            else PG_DMA(m0w + hop, voffW + hop, gb, (pw & 3) * 1024);
        };
        auto advance = [&]() __attribute__((always_inline)) {
            if (ti != z0 && ti != z1) {
                ++tr;
                gb += GG::B_ST;
                ga = (tr == nc0) ? ga1 : ga + GG::A_IMG;
            }
            ++ti;
        };
        auto stage = [&](int p, auto steady_t) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady_t)::value;
            // the pair's second chunk is virtual: the lane groups that contract it get zero A fragments (one AND per fragment register; a second
            // copy of the stage behind a branch made the register allocator spill the accumulators)
            const unsigned zm = ((2 * p + 1 == z0 || 2 * p + 1 == z1) && cs) ? 0u : ~0u;
            const unsigned sb = lds_base + (unsigned)(p & 1) * 2 * STAGE, sbn = lds_base + (unsigned)((p + 1) & 1) * 2 * STAGE;
            const bool has_next = STEADY || p + 1 < npair;
            if (2 * p == nv0 && nc1 > 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float f = s_fac[wm2 * 64 + 16 * i + l15];                     // row 16 i + l15 of the wave's 64
#pragma unroll
                    for (int j = 0; j < 7; ++j) c16[i][j] *= f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            auto gap = [&](int m) __attribute__((always_inline)) {
                const int pass = m / 21, gi = m % 21;
                // fragment reads
                if (pass < 3) {
                    if (gi == 0) PG_READ(fa[(pass + 1) & 1][0], sb + oAh, (pass + 1) * 1024);
                    if (gi == 1) PG_READ(fa[(pass + 1) & 1][1], sb + oAl, (pass + 1) * 1024);
                } else if (has_next) {
                    if (gi == 0) PG_READ(fa[0][0], sbn + oAh, 0);
                    if (gi == 1) PG_READ(fa[0][1], sbn + oAl, 0);
                    if (gi >= 1 && gi <= 7) PG_READ(fb[gi - 1][1], sbn + oBl, (gi - 1) * 1024);
                    if (gi >= 15 && gi <= 20) PG_READ(fb[gi - 15][0], sbn + oBh, (gi - 15) * 1024);
                    if (gi == 20) PG_READ(fb[6][0], sbn + oBh, 6 * 1024);
                }
                // DMA pieces (issuing waves): pair p + 2's first 7 in the read-free gaps of pass 3, pair p + 1's other 11 in passes 0 / 1
                if (ISS) {
                    if (pass == 3 && gi >= 8 && gi <= 14) {
                        if (STEADY || ti < nvt) piece(gi - 8);
                    }
                    if (pass == 0 && gi % 3 == 2) {                     // gaps 2, 5, .., 20: q = 7 .. 13
                        const int q = 7 + gi / 3;
                        if (STEADY || ti < nvt) {
                            piece(q < 9 ? q : q - 9);
                            if (q == 8) advance();
                        }
                    }
                    if (pass == 1 && gi % 3 == 2 && gi <= 11) {         // q = 14 .. 17
                        const int q = 14 + gi / 3;
                        if (STEADY || ti < nvt) {
                            piece(q - 9);
                            if (q == 17) advance();
                        }
                    }
                }
                if (pass == 2 && gi == 17) {                            // the pair's barrier: pair p + 1 has landed, pair p's slots are free behind it
                    if (ISS && has_next) PG_VMCNT(0);
                    __builtin_amdgcn_s_barrier();
                }
            };
    // (the WEIGHT fragment is the MFMA's first operand: a lane then owns four consecutive output COLUMNS 16 j + 4 g .. + 3 of token row 16 i + l15
    // -- the layout the epilogue works in -- instead of four rows of one column, and no transposition through LDS is needed)
#define PG_MFMA16(X, Y, m)                                                              \
    c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Y, X, c16[i][j], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    gap(m);                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // pass i: A tile i in fa[i & 1]; the weights' lo fragments landed long ago, the hi ones of a new pair need the 7-deep wait
                if (i == 0) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 a_h = fa[i & 1][0] & zm, a_l = fa[i & 1][1] & zm;
                const f16x8 ah = __builtin_bit_cast(f16x8, a_h), al = __builtin_bit_cast(f16x8, a_l);
#pragma unroll
                for (int j = 0; j < 7; ++j) { PG_MFMA16(ah, __builtin_bit_cast(f16x8, fb[j][1]), 21 * i + j) }
                if (i == 0) { asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int j = 0; j < 7; ++j) { PG_MFMA16(al, __builtin_bit_cast(f16x8, fb[j][0]), 21 * i + 7 + j) }
#pragma unroll
                for (int j = 0; j < 7; ++j) { PG_MFMA16(ah, __builtin_bit_cast(f16x8, fb[j][0]), 21 * i + 14 + j) }
            }
#undef PG_MFMA16
        };
        // prologue: pair 0 in flight | the epilogue's inputs fetched into LDS behind it | pair 0 landed | the first 7 pieces of pair 1
        if (ISS) {
#pragma unroll
            for (int q = 0; q < 9; ++q) piece(q);
            advance();
#pragma unroll
            for (int q = 0; q < 9; ++q) piece(q);
            advance();
            __builtin_amdgcn_sched_barrier(0);
            stage_inputs();
            __builtin_amdgcn_sched_barrier(0);
            PG_VMCNT(0);
#pragma unroll
            for (int q = 0; q < 7; ++q) piece(q);
        } else {
            stage_inputs();
        }
        __builtin_amdgcn_s_barrier();
        PG_READ(fa[0][0], lds_base + oAh, 0);
        PG_READ(fa[0][1], lds_base + oAl, 0);
#pragma unroll
        for (int j = 0; j < 7; ++j) PG_READ(fb[j][1], lds_base + oBl, j * 1024);
#pragma unroll
        for (int j = 0; j < 7; ++j) PG_READ(fb[j][0], lds_base + oBh, j * 1024);
        int p = 0;
        for (; p < npair - 2; ++p) stage(p, std::true_type{});
        for (; p < npair; ++p) stage(p, std::false_type{});
    };
    if constexpr (M16) {
        if (w < 4) run16(std::true_type{});
        else run16(std::false_type{});
    } else {
        if (wn == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 1>{});
    }
    if (G.dbg & 1) {                                             // timing builds: the main loop alone (the result must stay alive)
        float keep = 0.f;
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) keep += c16[i][j][0] + c16[i][j][1] + c16[i][j][2] + c16[i][j][3];
        } else {
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[j][r];
        }
        if (keep == 123.456f && P.out) P.out[0] = keep;
        return;
    }
    __syncthreads();                                             // every wave is done with the ring: the epilogue reuses it

    // ---- KSPLIT: the two workgroups of a row block swap half of their partial sums.  A wave's 32 rows are two rounds of 16 (accumulator
    // registers 0..7 / 8..15 of every tile): workgroup `khalf` KEEPS round khalf and gives the other round to its partner, as raw lane
    // registers (the partner's wave w, same lane, holds the same (row, column): identical tiling), 16 bytes per lane and store, sc1 both ways.
    // own + partner's is one fp32 addition whichever side does it: the result does not depend on which workgroup finishes a row.
    int keep = -1;                                               // the round this workgroup finishes (-1: both)
    if constexpr (KS_OK) {
        if (ksplit) {
            keep = khalf;
            const int give = 1 - khalf;
            int* const s_bad = reinterpret_cast<int*>(s_sum);   // (LDS word, free until the LayerNorm partials)
            if (t == 0) *s_bad = 0;
            float* const xsend = P.xk_buf + ((((size_t)rb * 2 + give) * 4 + w) * TNW * 2) * 256 + lane * 4;
            const float* const xrecv = P.xk_buf + ((((size_t)rb * 2 + khalf) * 4 + w) * TNW * 2) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const f32x4 x = {acc[j][8 * give + 4 * q2], acc[j][8 * give + 4 * q2 + 1], acc[j][8 * give + 4 * q2 + 2], acc[j][8 * give + 4 * q2 + 3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(xsend + (j * 2 + q2) * 256), "v"(x) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // every store of the workgroup has left
            unsigned* const fl = P.xk_flags + (size_t)rb * 2;
            if (t == 0) __hip_atomic_store(fl + khalf, P.xk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == 0) {                                        // bounded: a partner that never arrives poisons the rows and raises the status word
                unsigned spins = 0;
                while (__hip_atomic_load(fl + give, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.xk_epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {
                        if (lane == 0) { *s_bad = 1; if (P.xk_status) atomicOr(P.xk_status, 2u); }
                        break;
                    }
                }
            }
            __syncthreads();
            const bool bad = *s_bad != 0;
            f32x4 y[TNW][2];
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(y[j][q2]) : "v"(xrecv + (j * 2 + q2) * 256) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float poison = bad ? __uint_as_float(0x7fc00000u) : 0.f;
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][8 * khalf + 4 * q2 + e] += y[j][q2][e] + poison;
            __syncthreads();                                     // (s_bad lives in s_sum: read by everybody before the LayerNorm partials overwrite it)
            if (G.dbg & 64) {                                    // timing builds: main loop + exchange
                float kp = 0.f;
#pragma unroll
                for (int j = 0; j < TNW; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) kp += acc[j][r];
                if (kp == 123.456f && P.out) P.out[0] = kp;
                return;
            }
        }
    }
    auto skip = [&](int rr) __attribute__((always_inline)) { return KS_OK && keep >= 0 && rr != keep; };

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // The MFMA result has a lane's 16 values in 16 different rows.  Each wave transposes its 32 x 224 strip in two rounds of
    // 16 rows through a private [16][EP_S] float region; afterwards lane (lr = lane / 4, q = lane % 4) owns the float4s
    // 16 i + 4 q (i = 0 .. 13) of row 16 round + lr: a row is in ONE lane quad, so row statistics are two DPP steps.
    // M16: a wave holds 64 rows x 112 columns as 4 x 7 tiles of 16 x 16, and -- the weights being the MFMA's first operand -- lane (l15, g) holds
    // columns 4 g .. 4 g + 3 of row l15 of every tile: the accumulators ARE in the epilogue's layout (row lr = l15, float4 q = g of piece i = tile
    // column j; FOUR rounds = row tiles of 7 pieces).  No transposition; a row's four lanes are 16 apart (v_permlane16/32_swap instead of DPP).
    const int lr = M16 ? (lane & 15) : (lane >> 2), q = M16 ? (lane >> 4) : (lane & 3);
    constexpr int NR = M16 ? 4 : 2, NIE = M16 ? 7 : NI, EPS = M16 ? 112 : EP_S;
    const int wcol0 = M16 ? (w & 3) * 112 : wn * BNW, wrow0 = M16 ? (w >> 2) * 64 : wm * 32, wcw = M16 ? (w & 3) : wn;
    const int lrow = lr;
    float* const ep = reinterpret_cast<float*>(lds) + w * (16 * EPS);
    float cv[M16 ? 1 : TNW];                                     // 2^-s_c of the lane's accumulator columns (M16: read per piece)
    if constexpr (!M16) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) cv[j] = s_cinv[wn * BNW + l31 + 32 * j];
    }
    auto transpose_round = [&](int rr, float4 (&dst)[NIE]) __attribute__((always_inline)) {
        if constexpr (M16) {
            const float rinv = s_rinv[wrow0 + 16 * rr + lrow];
#pragma unroll
            for (int i = 0; i < NIE; ++i) {
                const float4 c4 = *reinterpret_cast<const float4*>(s_cinv + wcol0 + 16 * i + 4 * q);
                const f32x4 a = c16[rr][i];
                dst[i] = make_float4(a[0] * c4.x * rinv, a[1] * c4.y * rinv, a[2] * c4.z * rinv, a[3] * c4.w * rinv);
            }
            return;
        } else {
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = 8 * rr + r8;
                    ep[((r & 3) + 8 * ((r >> 2) & 1) + 4 * h) * EPS + 32 * j + l31] = acc[j][r] * cv[j];
                }
        }
        wave_fence();
        const float rinv = s_rinv[wrow0 + 16 * rr + lrow];       // undo the operand scales: exact powers of two
#pragma unroll
        for (int i = 0; i < NIE; ++i) {
            float4 x = *reinterpret_cast<const float4*>(ep + lr * EPS + 16 * i + 4 * q);
            x.x *= rinv; x.y *= rinv; x.z *= rinv; x.w *= rinv;
            dst[i] = x;
        }
        wave_fence();
    };
    const int colw = wcol0 + 4 * q;                              // first column of piece 0 inside the block
    auto add_bias = [&](float4 (&dst)[NIE]) __attribute__((always_inline)) {
        if (has_bias) {
#pragma unroll
            for (int i = 0; i < NIE; ++i) {
                const float4 b4 = *reinterpret_cast<const float4*>(s_bias + colw + 16 * i);
                dst[i].x += b4.x; dst[i].y += b4.y; dst[i].z += b4.z; dst[i].w += b4.w;
            }
        }
    };
    int grow[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) grow[rr] = rb * BM + wrow0 + 16 * rr + lrow;
    constexpr int mode = MODE;                                  // (one instantiation per epilogue: each gets its own register allocation)

    float4 v[NR][NIE];
    // sum over the four lanes that hold one row: a quad (DPP), or lanes 16 apart (M16: two lane-row swaps)
    auto row_sum4 = [&](float s) __attribute__((always_inline)) {
        if constexpr (M16) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const unsigned u = __float_as_uint(s);
            const u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // (rows 0 0 2 2), (rows 1 1 3 3)
            const float t2 = __uint_as_float(a.x) + __uint_as_float(a.y);
            const unsigned w2 = __float_as_uint(t2);
            const u32x2 b = __builtin_amdgcn_permlane32_swap(w2, w2, false, false);     // (low low), (high high)
            return __uint_as_float(b.x) + __uint_as_float(b.y);
        } else {
            s += dpp_xor1f(s);
            s += dpp_xor2f(s);
            return s;
        }
    };
    if (mode != PG_LN) {
        const int halfC = P.rot_C >> 1, rpad = P.rot_piece_pad, rlen = P.rot_piece_len;
        const unsigned rmagic = rpad > 0 ? ((1u << 20) + rpad - 1) / rpad : 0u;   // col / rpad = (col * rmagic) >> 20 for col < 4096, rpad < 1024
        const float scale = P.scale;
        // Every rotary table load precedes the wave's first store: vmcnt retires in order, so a load behind a store would wait
        // for the store to reach memory (loading the tables piece by piece between the stores cost 37 us on the q|k|v launch).
        // Order: tables 0 | round 0 -> rotated in place | tables 1 | round 1 | all stores.
        // Head-padded outputs (rot_piece_pad > 0: column c' = pad (c / len) + c % len of the image is column c of the nn.Linear):
        // the table index follows the nn.Linear's column.
        // the table entries of piece i of a row: (cos_k, cos_k+1, sin_k, sin_k+1)
#define PG_TABLE_INDEX(i)                                                                                                   \
    int col = min(colw + 16 * (i), C - 4);                                                                                  \
    if (rpad > 0) { const int hq = (int)(((unsigned)col * rmagic) >> 20); col = hq * rlen + min(col - hq * rpad, rlen - 4); } \
    if (col >= P.rot_C) col %= P.rot_C; /* (rot_C = C in every caller: the division is never reached) */                    \
    const int ridx = col >> 1;
#define PG_LOAD_TABLES(rr, tb)                                                                                              \
    {                                                                                                                       \
    if (P.csT) { /* one 16-byte load per piece: (cos_k, sin_k, cos_k+1, sin_k+1) */                                         \
        const float* pp = P.csT + (size_t)min(grow[rr], rows - 1) * halfC * 2;                                              \
        _Pragma("unroll") for (int i = 0; i < NIE; ++i) {                                                                   \
            PG_TABLE_INDEX(i)                                                                                               \
            const float4 cs = *reinterpret_cast<const float4*>(pp + 2 * ridx);                                              \
            tb[i] = make_float4(cs.x, cs.z, cs.y, cs.w);                                                                    \
        }                                                                                                                   \
    } else {                                                                                                                \
        const float* cp = P.cosT + (size_t)min(grow[rr], rows - 1) * halfC;                                                 \
        const float* sp = P.sinT + (size_t)min(grow[rr], rows - 1) * halfC;                                                 \
        _Pragma("unroll") for (int i = 0; i < NIE; ++i) {                                                                   \
            PG_TABLE_INDEX(i)                                                                                               \
            const float2 c = *reinterpret_cast<const float2*>(cp + ridx), sn = *reinterpret_cast<const float2*>(sp + ridx); \
            tb[i] = make_float4(c.x, c.y, sn.x, sn.y);                                                                      \
        }                                                                                                                   \
    }                                                                                                                       \
    }
        auto finish_round = [&](int rr, const float4 (&tb)[NIE]) __attribute__((always_inline)) {
            add_bias(v[rr]);
#pragma unroll
            for (int i = 0; i < NIE; ++i) {
                float4 x = v[rr][i];
                if (rot) {
                    // x cos + swap(x) sin, swap(x)[2k] = -x[2k+1], swap(x)[2k+1] = x[2k]  (position_encoding.py:25-35)
                    const float x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
                    x.x = __fadd_rn(__fmul_rn(x0, tb[i].x), __fmul_rn(-x1, tb[i].z));
                    x.y = __fadd_rn(__fmul_rn(x1, tb[i].x), __fmul_rn(x0, tb[i].z));
                    x.z = __fadd_rn(__fmul_rn(x2, tb[i].y), __fmul_rn(-x3, tb[i].w));
                    x.w = __fadd_rn(__fmul_rn(x3, tb[i].y), __fmul_rn(x2, tb[i].w));
                }
                x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale;
                if (mode == PG_PLANES && P.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
                v[rr][i] = x;
            }
        };
        if constexpr (!M16) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                float4 tb[NIE];
                if (rot && !(G.dbg & 32)) PG_LOAD_TABLES(rr, tb)
                transpose_round(rr, v[rr]);
                finish_round(rr, tb);
                __builtin_amdgcn_sched_barrier(0);               // keep the stores below behind the loads of the next round
            }
        } else {
            // four rounds of 7 pieces: the tables of TWO rounds are in flight while those two rounds are transposed, i.e. two exposed load
            // latencies per workgroup as in the two-round form: tables 0, 1 | rounds 0, 1 | rotate 0 | tables 2 | rotate 1 | tables 3 | ..
            float4 tb0[NIE], tb1[NIE];
            if (rot && !(G.dbg & 32)) { PG_LOAD_TABLES(0, tb0) PG_LOAD_TABLES(1, tb1) }
            transpose_round(0, v[0]);
            transpose_round(1, v[1]);
            finish_round(0, tb0);
            __builtin_amdgcn_sched_barrier(0);
            if (rot && !(G.dbg & 32)) PG_LOAD_TABLES(2, tb0)
            finish_round(1, tb1);
            __builtin_amdgcn_sched_barrier(0);
            if (rot && !(G.dbg & 32)) PG_LOAD_TABLES(3, tb1)
            transpose_round(2, v[2]);
            transpose_round(3, v[3]);
            finish_round(2, tb0);
            finish_round(3, tb1);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef PG_LOAD_TABLES
#undef PG_TABLE_INDEX
        if (mode == PG_F32) {
            float* __restrict__ outp = P.out + (size_t)nb * P.blk_stride;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if (grow[rr] >= rows) continue;
#pragma unroll
                for (int i = 0; i < NIE; ++i) {
                    const int col = colw + 16 * i;
                    if (wcol0 + 16 * i < C && !(G.dbg & 4)) *reinterpret_cast<float4*>(outp + (size_t)grow[rr] * P.ldo + col) = v[rr][i];
                }
            }
            return;
        }
        if (mode == PG_PLANES && P.out) {                        // optional fp32 copy of the block (the residual stream of a consumer)
            float* __restrict__ outp = P.out + (size_t)nb * P.blk_stride;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if (grow[rr] >= rows) continue;
#pragma unroll
                for (int i = 0; i < NIE; ++i) {
                    const int col = colw + 16 * i;
                    if (wcol0 + 16 * i < C && !(G.dbg & 4)) *reinterpret_cast<float4*>(outp + (size_t)grow[rr] * P.ldo + col) = v[rr][i];
                }
            }
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) if (!skip(rr)) transpose_round(rr, v[rr]);      // (KSPLIT: only the round this workgroup finishes)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) if (!skip(rr)) add_bias(v[rr]);
    }

    if (mode == PG_LN) {
        // nn.LayerNorm over the C columns of the block (biased variance, eps inside the sqrt; transformero.py:88-94)
        // Residual rows: EVERY load is issued here, before the first store of this wave -- vmcnt retires in order, so a load
        // behind a store waits for the store to reach memory (the interleaved form cost 40 us per launch).  The first half of the rounds
        // stays in registers, the other half is parked in LDS (the wave's transposition region, free now; M16: a second region behind the eight).
        const float* __restrict__ res = P.resid;
        constexpr int NRR = NR / 2;                              // rounds whose residual rows stay in registers
        float4 r0[NRR][NIE];
        auto park = [&](int rr) __attribute__((always_inline)) { return ep + (rr - NRR) * (8 * 16 * EPS) + lr * EPS + 4 * q; };
        if (res && !(G.dbg & 16)) {
#pragma unroll
            for (int rr = NRR; rr < NR; ++rr) {
                if (skip(rr)) continue;
                const float* rp = res + (size_t)min(grow[rr], rows - 1) * P.ldr;
#pragma unroll
                for (int i = 0; i < NIE; ++i) {
                    const int col = colw + 16 * i;
                    const float4 x = wcol0 + 16 * i < C ? *reinterpret_cast<const float4*>(rp + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(park(rr) + 16 * i) = x;
                }
            }
#pragma unroll
            for (int rr = 0; rr < NRR; ++rr) {
                if (skip(rr)) continue;
                const float* rp = res + (size_t)min(grow[rr], rows - 1) * P.ldr;
#pragma unroll
                for (int i = 0; i < NIE; ++i) {
                    const int col = colw + 16 * i;
                    r0[rr][i] = wcol0 + 16 * i < C ? *reinterpret_cast<const float4*>(rp + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        const bool postadd = res && P.ln_postadd;                // LayerNorm(acc + resid): the residual joins BEFORE the statistics
        if (postadd) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int i = 0; i < NIE; ++i) {
                    if (skip(rr)) continue;
                    const float4 r4 = rr < NRR ? r0[rr < NRR ? rr : 0][i] : *reinterpret_cast<const float4*>(park(rr) + 16 * i);
                    v[rr][i].x += r4.x; v[rr][i].y += r4.y; v[rr][i].z += r4.z; v[rr][i].w += r4.w;
                }
        }
        float mean[NR], rstd[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (skip(rr)) continue;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NIE; ++i)
                if (wcol0 + 16 * i < C) s += (v[rr][i].x + v[rr][i].y) + (v[rr][i].z + v[rr][i].w);
            s = row_sum4(s);
            if (q == 0) s_sum[wcw * 128 + wrow0 + 16 * rr + lrow] = s;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (skip(rr)) continue;
            const int rl = wrow0 + 16 * rr + lrow;
            float tot = s_sum[rl];
#pragma unroll
            for (int c = 1; c < NCW; ++c) tot += s_sum[c * 128 + rl];
            mean[rr] = tot / (float)C;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NIE; ++i)
                if (wcol0 + 16 * i < C) {
                    const float d0 = v[rr][i].x - mean[rr], d1 = v[rr][i].y - mean[rr], d2 = v[rr][i].z - mean[rr], d3 = v[rr][i].w - mean[rr];
                    s = fmaf(d0, d0, s); s = fmaf(d1, d1, s); s = fmaf(d2, d2, s); s = fmaf(d3, d3, s);
                }
            s = row_sum4(s);
            if (q == 0) s_sq[wcw * 128 + rl] = s;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (skip(rr)) continue;
            const int rl = wrow0 + 16 * rr + lrow;
            float tot = s_sq[rl];
#pragma unroll
            for (int c = 1; c < NCW; ++c) tot += s_sq[c * 128 + rl];
            rstd[rr] = 1.0f / sqrtf(tot / (float)C + 1e-5f);
            const bool rok = grow[rr] < rows;
#pragma unroll
            for (int i = 0; i < NIE; ++i) {
                const int col = colw + 16 * i;
                if (wcol0 + 16 * i < C) {                        // (C % 16 == 0: a 16-column piece is inside or outside as a whole -- a wave-uniform test)
                    const float4 g4 = *reinterpret_cast<const float4*>(s_gam + col), b4 = *reinterpret_cast<const float4*>(s_bet + col);
                    float4 y;
                    y.x = (v[rr][i].x - mean[rr]) * rstd[rr] * g4.x + b4.x; y.y = (v[rr][i].y - mean[rr]) * rstd[rr] * g4.y + b4.y;
                    y.z = (v[rr][i].z - mean[rr]) * rstd[rr] * g4.z + b4.z; y.w = (v[rr][i].w - mean[rr]) * rstd[rr] * g4.w + b4.w;
                    if (res && !postadd) {
                        const float4 r4 = rr < NRR ? r0[rr < NRR ? rr : 0][i] : *reinterpret_cast<const float4*>(park(rr) + 16 * i);
                        y.x += r4.x; y.y += r4.y; y.z += r4.z; y.w += r4.w;
                    }
                    v[rr][i] = y;
                    if (P.out && rok && !(G.dbg & 4)) *reinterpret_cast<float4*>(P.out + (size_t)grow[rr] * P.ldo + col) = y;
                }
            }
        }
    }
    if (!P.pimg) return;

    // ---- plane image of the result.  Piece i of lane q is columns 16 i + 4 q .. + 3 of chunk (wn BNW / 16 + i): the lanes
    // q and q ^ 1 hold the two halves of one hi unit and of one lo unit (16 bytes each); the even lane stores the hi unit, the odd one the lo unit.
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (skip(rr)) continue;
        const bool rok = grow[rr] < rows;
        const float bound = s_bound[wrow0 + 16 * rr + lrow];
        if (P.pbnd && rok && (nb == 0 || per_blk) && wcw == 0 && q == 0) P.pbnd[(size_t)nb * P.pbnd_blk_stride + grow[rr]] = bound;
        const float sc = pow2i(scale_exp(bound));
        const int rl = sub + wrow0 + 16 * rr + lrow, swz = (rl >> 2) & 3;       // row inside the 128-row image block
        char* const rowp = P.pimg + (size_t)nb * P.pimg_blk_stride +
                           (((size_t)rb128 * P.p_nct + P.p_kc0 + (per_blk ? 0 : nb * (C >> 4)) + (wcol0 >> 4)) * 128 + rl) * 64;
        const unsigned uh = (unsigned)(((q >> 1) ^ swz) << 4), ul = (unsigned)(((2 + (q >> 1)) ^ swz) << 4);
#pragma unroll
        for (int i = 0; i < NIE; ++i) {
            if (wcol0 + 16 * i >= C) continue;
            unsigned h0, l0, h1, l1;
            split2(v[rr][i].x * sc, v[rr][i].y * sc, h0, l0);
            split2(v[rr][i].z * sc, v[rr][i].w * sc, h1, l1);
            // ONE 16-byte store per lane: the even lane of a pair writes the hi unit (its 4 columns, then the partner's), the odd lane the lo
            // unit -- every lane stores, a wave instruction covers 16 rows x 64 contiguous bytes (two stores by half the lanes before:
            // 39 of the family's 108 ms per pass were these stores, measured by ablation in round 2)
            const bool odd = q & 1;
            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
            u32x4v U;
            if constexpr (M16) {
                // the partner is 16 lanes away: v_permlane16_swap exchanges the odd lane rows of its first operand with the even ones of its second --
                // the even lane ends with (own hi, partner's hi), the odd one with (partner's lo, own lo): both store {x0, x1, y0, y1}
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 e0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false), e1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
                U = u32x4v{e0.x, e1.x, e0.y, e1.y};
            } else {
                const unsigned r0_ = dpp_xor1(odd ? h0 : l0), r1_ = dpp_xor1(odd ? h1 : l1);  // what the partner stores of mine <-> what I store of the partner's
                U = odd ? u32x4v{r0_, r1_, l0, l1} : u32x4v{h0, h1, r0_, r1_};
            }
            if (rok && !(G.dbg & 8)) __builtin_nontemporal_store(U, reinterpret_cast<u32x4v*>(rowp + (size_t)i * 8192 + (odd ? ul : uh)));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The WIDE-WAVE geometry (round 6; the 576-column blocks of 4DMatch, C = 528 / 4 heads padded to 144): 128 rows x 288 columns per workgroup,
// FOUR waves as 2 (rows) x 2 (columns) of 64 x 144 = 4 x 9 tiles of v_mfma_f32_16x16x32_f16, ONE wave per SIMD (144 accumulator + 72 weight-
// fragment registers).  Why: the 64-row workgroups of the 576-column kernel stage 4 KB of A + 36 KB of weights per 16-deep k-chunk and run at
// what a CU can take in (50-56 GB/s measured: merge / mlp0 / qkv main loops at 0.71-0.85 us per chunk against 0.36 us of MFMA time); a 128 x 288
// tile has the same area and stages 8 + 18 KB.  A logical column block of up to 576 columns is TWO physical sub-blocks of 288 weight rows
// (PgW::sub = 2: image [nblk][2][nct][288 rows][64 B]; cinv / wnorm stay logical), dealt like column blocks.  The main loop is run16's (chunk
// pairs, a 4-slot ring, one barrier per pair, weight fragments loaded once per pair), every wave issuing its share of the pair's 52 DMA pieces
// (13-14 per wave) in the read-free gaps of pass 3 and in pass 0.  Epilogues: PG_F32 and PG_PLANES (the launches without LayerNorm: q | k | v,
// mlp0, the matching head's projection); the accumulators are in the epilogue's layout as in the 16x16x32 form above.
// ---------------------------------------------------------------------------------------------------------------------
struct PgGeomW {
    static constexpr int NJ = 9, BN = 288, LBN = 576, BM = 128, NTHR = 256, A_ST = BM * 64, B_ST = BN * 64, STAGE = A_ST + B_ST, NST = 4;
    static constexpr int RING = NST * STAGE;
    // + fac[128], rinv[128], red[16], bias[BN], cinv[BN], bound[128]
    static constexpr int SMEM = RING + (128 + 128 + 16 + BN + BN + 128) * 4;
};

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void pgemm16w_kernel(PgBatch G) {
    using GG = PgGeomW;
    constexpr int NJ = GG::NJ, BN = GG::BN, STAGE = GG::STAGE, A_ST = GG::A_ST, BM = GG::BM, NTHR = GG::NTHR, NST = GG::NST, NG = 3 * NJ;
    static_assert(MODE == PG_F32 || MODE == PG_PLANES, "the wide-wave geometry has no LayerNorm epilogue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;

    const PgProblem& P = G.p[blockIdx.y];
    // KSPLIT (launches of at most half a chip of tiles: cfg3's one-sided mlp0): a tile's k range is split over TWO workgroups, ids 8 apart (one XCD) --
    // a two-segment operand by segment ([x | msg]: workgroup 0 contracts x, workgroup 1 msg), a one-segment operand at an even chunk -- which swap two of a
    // wave's four rounds of partial sums behind the main loop (see pgemm_kernel's KSPLIT) and each run half of the epilogue
    const bool ksplit = P.ksplit != 0;
    const int rows = P.rows, C = P.C, nblk = P.nblk, nc0f = P.nc0, nc1f = P.A1 ? P.nc1 : 0, nstf = nc0f + nc1f;
    const int rbs = (rows + BM - 1) / BM, nb2 = 2 * nblk * (ksplit ? 2 : 1);
    const int grp = blockIdx.x / (8 * nb2), rem = blockIdx.x % (8 * nb2);
    const int rb = grp * 8 + (rem & 7), tix = rem >> 3, khalf = ksplit ? (tix & 1) : 0, pb = ksplit ? (tix >> 1) : tix, nb = pb >> 1, csub = (pb & 1) * BN;
    if (rb >= rbs || csub >= C) return;
    // this workgroup's chunks: [c_first, c_first + nc0) of the weight image, from A image `Aimg` (a_chunks chunks per row block) starting at chunk a_first
    int nc0 = nc0f, nc1 = nc1f, c_first = 0, a_first = 0, a_chunks = nc0f;
    const char* Aimg = P.A0;
    bool to_seg1_scale = false;                                 // the accumulators of segment 0 are brought to segment 1's row scale behind the loop
    if (ksplit) {
        if (nc1f > 0) {
            nc1 = 0;
            if (khalf == 0) to_seg1_scale = true;
            else { Aimg = P.A1; a_chunks = nc1f; c_first = nc0f; nc0 = nc1f; }
        } else {
            const int kb = (nc0f / 2) & ~1;
            if (khalf == 0) nc0 = kb;
            else { a_first = c_first = kb; nc0 = nc0f - kb; }
        }
    }
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wn = w & 1, wm2 = w >> 1;

    float* const s_fac = reinterpret_cast<float*>(lds + GG::RING);
    float* const s_rinv = s_fac + 128;
    float* const s_red = s_rinv + 128;
    float* const s_bias = s_red + 16;
    float* const s_cinv = s_bias + BN;
    float* const s_bound = s_cinv + BN;
    const bool has_bias = P.bias != nullptr;
    const bool rot = (P.rot_mask >> nb) & 1;
    const bool per_blk = P.pimg_blk_stride != 0;
    auto stage_inputs = [&]() __attribute__((always_inline)) {
        if (has_bias) {
            const float* bp = P.bias + (size_t)nb * C + csub;
            for (int c = t; c < BN; c += NTHR) s_bias[c] = csub + c < C ? bp[c] : 0.f;
        }
        {
            const float* cp = P.W.cinv + (size_t)nb * GG::LBN + csub;
            for (int c = t; c < BN; c += NTHR) s_cinv[c] = cp[c];
        }
        float gmax = 0.f;
        if (((P.grp_mask >> nb) & 1) && !P.grp_bnd) {
            const int gbase = (rb * BM) / P.grp_rows * P.grp_rows;
            float m = 0.f;
            for (int i = t; i < P.grp_rows; i += NTHR) m = fmaxf(m, P.bnd0[gbase + i]);
            m = wave_max(m);
            if (lane == 0) s_red[w] = m;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NTHR / 64; ++k) gmax = fmaxf(gmax, s_red[k]);
        }
        if (t < BM) {
            const int row = min(rb * BM + t, rows - 1);
            const float b0 = P.bnd0[row], b1 = nc1f > 0 ? P.bnd1[row] : 0.f;
            const int e0 = scale_exp(b0);
            int e1 = e0;
            if (nc1f > 0) e1 = scale_exp(b1);
            s_fac[t] = pow2i(min(max(e1 - e0, -120), 120));
            s_rinv[t] = pow2i(-e1);
            if (P.pimg) {
                // (the bound rules of pgemm_kernel, word for word: both kernels must scale a row of one image alike)
                const float bin = ((P.grp_mask >> nb) & 1) ? (P.grp_bnd ? P.grp_bnd[P.grp_first + row / P.grp_rows] : gmax) : fmaxf(b0, b1);
                float wn_ = P.W.wnorm[nb];
                if (!per_blk)
                    for (int b2 = 0; b2 < nblk; ++b2) wn_ = fmaxf(wn_, P.W.wnorm[b2]);
                float bm = 0.f;
                if (P.bias_max) {
                    bm = P.bias_max[nb];
                    if (!per_blk)
                        for (int b2 = 0; b2 < nblk; ++b2) bm = fmaxf(bm, P.bias_max[b2]);
                }
                s_bound[t] = (bin * wn_ + bm) * (rot ? 1.41421366f : 1.f) * fabsf(P.scale);
            }
        }
    };

    // ---- main loop (see run16): lane group g = lane / 16 reads unit (g >> 1) of chunk c + (g & 1)
    const int l15 = lane & 15, g = lane >> 4, cs = g & 1, uh = g >> 1, sw16 = (l15 >> 2) & 3;
    const unsigned oAh = cs * STAGE + (wm2 * 64 + l15) * 64 + ((uh ^ sw16) << 4), oAl = cs * STAGE + (wm2 * 64 + l15) * 64 + (((2 + uh) ^ sw16) << 4);
    const unsigned oBh = cs * STAGE + A_ST + (wn * 144 + l15) * 64 + ((uh ^ sw16) << 4), oBl = cs * STAGE + A_ST + (wn * 144 + l15) * 64 + (((2 + uh) ^ sw16) << 4);
    f32x4 c16[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) c16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 fb[NJ][2], fa[2][2];
    const int nv0 = (nc0 + 1) & ~1, nvt = nv0 + ((nc1 + 1) & ~1), npair = nvt >> 1;
    const int z0 = (nc0 & 1) ? nv0 - 1 : -1, z1 = (nc1 & 1) ? nvt - 1 : -1;
    // DMA pieces of a chunk: 8 of A (two per wave) + 18 of weights (waves 0, 1: five, waves 2, 3: four, contiguous)
    const int npw = w < 2 ? 5 : 4, st0 = w < 2 ? w * 5 : 10 + (w - 2) * 4;
    const unsigned voffW = lane * 16 + st0 * 1024, voffA = lane * 16 + 2 * w * 1024;
    const char* ga = Aimg + ((size_t)rb * a_chunks + a_first) * GG::A_ST;
    const char* gb = P.W.img + ((size_t)(nb * 2 + (pb & 1)) * nstf + c_first) * GG::B_ST;
    const char* const ga1 = nc1 > 0 ? P.A1 + (size_t)rb * nc1 * GG::A_ST : nullptr;
    int ti = 0, tr = 0;
    auto piece = [&](int q) __attribute__((always_inline)) {            // q = 0 .. 6 of chunk ti
        if (ti == z0 || ti == z1) return;
        const unsigned dstb = lds_base + (unsigned)(ti % NST) * STAGE;
        const unsigned m0w = dstb + A_ST + st0 * 1024;
        const int pw = q - 2;
        const unsigned hop = (unsigned)(pw >> 2) * 4096;
        if (q < 2) PG_DMA(dstb + 2 * w * 1024, voffA, ga, q * 1024);
        else if (pw < npw) PG_DMA(m0w + hop, voffW + hop, gb, (pw & 3) * 1024);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (ti != z0 && ti != z1) {
            ++tr;
            gb += GG::B_ST;
            ga = (tr == nc0) ? ga1 : ga + GG::A_ST;
        }
        ++ti;
    };
    auto stage = [&](int p, auto steady_t) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_t)::value;
        const unsigned zm = ((2 * p + 1 == z0 || 2 * p + 1 == z1) && cs) ? 0u : ~0u;
        const unsigned sb = lds_base + (unsigned)(p & 1) * 2 * STAGE, sbn = lds_base + (unsigned)((p + 1) & 1) * 2 * STAGE;
        const bool has_next = STEADY || p + 1 < npair;
        if (2 * p == nv0 && nc1 > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f = s_fac[wm2 * 64 + 16 * i + l15];
#pragma unroll
                for (int j = 0; j < NJ; ++j) c16[i][j] *= f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        auto gap = [&](int m) __attribute__((always_inline)) {
            const int pass = m / NG, gi = m % NG;
            if (pass < 3) {
                if (gi == 0) PG_READ(fa[(pass + 1) & 1][0], sb + oAh, (pass + 1) * 1024);
                if (gi == 1) PG_READ(fa[(pass + 1) & 1][1], sb + oAl, (pass + 1) * 1024);
            } else if (has_next) {
                if (gi == 0) PG_READ(fa[0][0], sbn + oAh, 0);
                if (gi == 1) PG_READ(fa[0][1], sbn + oAl, 0);
                if (gi >= 1 && gi <= NJ) PG_READ(fb[gi - 1][1], sbn + oBl, (gi - 1) * 1024);
                if (gi >= 2 * NJ + 1 && gi <= NG - 1) PG_READ(fb[gi - 2 * NJ - 1][0], sbn + oBh, (gi - 2 * NJ - 1) * 1024);
                if (gi == NG - 1) PG_READ(fb[NJ - 1][0], sbn + oBh, (NJ - 1) * 1024);
            }
            // DMA: the first chunk of pair p + 2 in the read-free gaps of pass 3 (its slot is free behind the barrier), the second chunk of
            // pair p + 1 in pass 0 (every third gap)
            if (pass == 3 && gi >= NJ + 1 && gi <= NJ + 7) {
                if (STEADY || ti < nvt) {
                    piece(gi - NJ - 1);
                    if (gi == NJ + 7) advance();
                }
            }
            if (pass == 0 && gi % 3 == 2 && gi <= 20) {
                if (STEADY || ti < nvt) {
                    piece(gi / 3);
                    if (gi == 20) advance();
                }
            }
            if (pass == 2 && gi == NG - 5) {                         // the pair's barrier: pair p + 1 has landed, pair p's slots are free behind it
                if (has_next) PG_VMCNT(0);
                __builtin_amdgcn_s_barrier();
            }
        };
#define PG_MFMA16W(X, Y, m)                                                             \
    c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Y, X, c16[i][j], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    gap(m);                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 0) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 a_h = fa[i & 1][0] & zm, a_l = fa[i & 1][1] & zm;
            const f16x8 ah = __builtin_bit_cast(f16x8, a_h), al = __builtin_bit_cast(f16x8, a_l);
#pragma unroll
            for (int j = 0; j < NJ; ++j) { PG_MFMA16W(ah, __builtin_bit_cast(f16x8, fb[j][1]), NG * i + j) }
            if (i == 0) { asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { PG_MFMA16W(al, __builtin_bit_cast(f16x8, fb[j][0]), NG * i + NJ + j) }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { PG_MFMA16W(ah, __builtin_bit_cast(f16x8, fb[j][0]), NG * i + 2 * NJ + j) }
        }
#undef PG_MFMA16W
    };
    // prologue: pair 0 in flight | the epilogue's inputs staged behind it | pair 0 landed | the first chunk of pair 1
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
        for (int q = 0; q < 7; ++q) piece(q);
        advance();
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_inputs();
    __builtin_amdgcn_sched_barrier(0);
    PG_VMCNT(0);
    if (ti < nvt) {
#pragma unroll
        for (int q = 0; q < 7; ++q) piece(q);
        advance();
    }
    __builtin_amdgcn_s_barrier();
    PG_READ(fa[0][0], lds_base + oAh, 0);
    PG_READ(fa[0][1], lds_base + oAl, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) PG_READ(fb[j][1], lds_base + oBl, j * 1024);
#pragma unroll
    for (int j = 0; j < NJ; ++j) PG_READ(fb[j][0], lds_base + oBh, j * 1024);
    {
        int p = 0;
        for (; p < npair - 2; ++p) stage(p, std::true_type{});
        for (; p < npair; ++p) stage(p, std::false_type{});
    }
    if (G.dbg & 1) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) keep += c16[i][j][0] + c16[i][j][1] + c16[i][j][2] + c16[i][j][3];
        if (keep == 123.456f && P.out) P.out[0] = keep;
        return;
    }
    __syncthreads();

    // ---- KSPLIT: bring a segment-0 partial to segment 1's row scale, then swap rounds: workgroup `khalf` keeps rounds 2 khalf, 2 khalf + 1 of every
    // wave and gives the other two to its partner (raw lane registers: identical tiling), sc1 both ways, epoch flags, bounded spin
    if (ksplit) {
        if (to_seg1_scale) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f = s_fac[wm2 * 64 + 16 * i + l15];
#pragma unroll
                for (int j = 0; j < NJ; ++j) c16[i][j] *= f;
            }
        }
        const int give = 1 - khalf;
        int* const s_bad = reinterpret_cast<int*>(s_red);
        if (t == 0) *s_bad = 0;
        const size_t tile = (size_t)rb * (2 * nblk) + pb;
        float* const xsend = P.xk_buf + (((tile * 2 + give) * 4 + w) * NJ * 2) * 256 + lane * 4;
        const float* const xrecv = P.xk_buf + (((tile * 2 + khalf) * 4 + w) * NJ * 2) * 256 + lane * 4;
        // (register arrays: constant indices only -- one copy of the loop per half)
        auto send = [&](auto GV) __attribute__((always_inline)) {
            constexpr int g0 = 2 * decltype(GV)::value;
            float* const xs = xsend;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const f32x4 x = c16[g0 + r2][j];
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(xs + (j * 2 + r2) * 256), "v"(x) : "memory");
                }
        };
        if (give == 0) send(std::integral_constant<int, 0>{}); else send(std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const fl = P.xk_flags + tile * 2;
        if (t == 0) __hip_atomic_store(fl + khalf, P.xk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(fl + give, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.xk_epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {
                    if (lane == 0) { *s_bad = 1; if (P.xk_status) atomicOr(P.xk_status, 2u); }
                    break;
                }
            }
        }
        __syncthreads();
        const float poison = *s_bad != 0 ? __uint_as_float(0x7fc00000u) : 0.f;
        // (all eighteen loads of a lane in flight before the first is waited for: one round trip, not eighteen)
        auto recv = [&](auto KV) __attribute__((always_inline)) {
            constexpr int k0 = 2 * decltype(KV)::value;
            const float* const xr = xrecv;
            f32x4 y[NJ][2];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(y[j][r2]) : "v"(xr + (j * 2 + r2) * 256) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) c16[k0 + r2][j] += y[j][r2] + poison;
        };
        if (khalf == 0) recv(std::integral_constant<int, 0>{}); else recv(std::integral_constant<int, 1>{});
        __syncthreads();
    }
    auto skip = [&](int rr) __attribute__((always_inline)) { return ksplit && (rr >> 1) != khalf; };

    // ---- epilogue: lane (l15, g) holds columns 4 g .. 4 g + 3 of row l15 of every 16 x 16 tile: row lr = l15, float4 q = g of piece i = tile column
    const int lr = l15, q = g;
    constexpr int NR = 4, NIE = NJ;
    const int wloc = wn * 144, wcol0 = csub + wloc, wrow0 = wm2 * 64;      // the wave's first column inside the sub-block / the logical block
    const int colw = wcol0 + 4 * q;
    int grow[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) grow[rr] = rb * BM + wrow0 + 16 * rr + lr;
    float4 v[NR][NIE];
    auto scale_round = [&](int rr, float4 (&dst)[NIE]) __attribute__((always_inline)) {
        const float rinv = s_rinv[wrow0 + 16 * rr + lr];
#pragma unroll
        for (int i = 0; i < NIE; ++i) {
            const float4 c4 = *reinterpret_cast<const float4*>(s_cinv + wloc + 16 * i + 4 * q);
            const f32x4 a = c16[rr][i];
            dst[i] = make_float4(a[0] * c4.x * rinv, a[1] * c4.y * rinv, a[2] * c4.z * rinv, a[3] * c4.w * rinv);
        }
    };
    const int halfC = P.rot_C >> 1, rpad = P.rot_piece_pad, rlen = P.rot_piece_len;
    const unsigned rmagic = rpad > 0 ? ((1u << 20) + rpad - 1) / rpad : 0u;
    const float scale = P.scale;
    auto load_tables = [&](int rr, float4 (&tb)[NIE]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIE; ++i) {
            int col = min(colw + 16 * i, C - 4);
            if (rpad > 0) { const int hq = (int)(((unsigned)col * rmagic) >> 20); col = hq * rlen + min(col - hq * rpad, rlen - 4); }
            if (col >= P.rot_C) col %= P.rot_C;
            const int ridx = col >> 1;
            if (P.csT) {
                const float4 cs4 = *reinterpret_cast<const float4*>(P.csT + (size_t)min(grow[rr], rows - 1) * halfC * 2 + 2 * ridx);
                tb[i] = make_float4(cs4.x, cs4.z, cs4.y, cs4.w);
            } else {
                const float2 c = *reinterpret_cast<const float2*>(P.cosT + (size_t)min(grow[rr], rows - 1) * halfC + ridx);
                const float2 sn = *reinterpret_cast<const float2*>(P.sinT + (size_t)min(grow[rr], rows - 1) * halfC + ridx);
                tb[i] = make_float4(c.x, c.y, sn.x, sn.y);
            }
        }
    };
    auto finish_round = [&](int rr, const float4 (&tb)[NIE]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIE; ++i) {
            float4 x = v[rr][i];
            if (has_bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(s_bias + wloc + 4 * q + 16 * i);
                x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
            }
            if (rot) {
                const float x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
                x.x = __fadd_rn(__fmul_rn(x0, tb[i].x), __fmul_rn(-x1, tb[i].z));
                x.y = __fadd_rn(__fmul_rn(x1, tb[i].x), __fmul_rn(x0, tb[i].z));
                x.z = __fadd_rn(__fmul_rn(x2, tb[i].y), __fmul_rn(-x3, tb[i].w));
                x.w = __fadd_rn(__fmul_rn(x3, tb[i].y), __fmul_rn(x2, tb[i].w));
            }
            x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale;
            if (MODE == PG_PLANES && P.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
            v[rr][i] = x;
        }
    };
    if (ksplit) {
        // (half of the rounds: the pair this workgroup keeps)
        float4 tb0[NIE], tb1[NIE];
        if (khalf == 0) {
            if (rot) { load_tables(0, tb0); load_tables(1, tb1); }
            scale_round(0, v[0]); scale_round(1, v[1]);
            finish_round(0, tb0); finish_round(1, tb1);
        } else {
            if (rot) { load_tables(2, tb0); load_tables(3, tb1); }
            scale_round(2, v[2]); scale_round(3, v[3]);
            finish_round(2, tb0); finish_round(3, tb1);
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
        // every rotary table load precedes the wave's first store (vmcnt retires in order): tables of two rounds in flight at a time
        float4 tb0[NIE], tb1[NIE];
        if (rot) { load_tables(0, tb0); load_tables(1, tb1); }
        scale_round(0, v[0]);
        scale_round(1, v[1]);
        finish_round(0, tb0);
        __builtin_amdgcn_sched_barrier(0);
        if (rot) load_tables(2, tb0);
        finish_round(1, tb1);
        __builtin_amdgcn_sched_barrier(0);
        if (rot) load_tables(3, tb1);
        scale_round(2, v[2]);
        scale_round(3, v[3]);
        finish_round(2, tb0);
        finish_round(3, tb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == PG_F32 || P.out) {
        float* __restrict__ outp = P.out + (size_t)nb * P.blk_stride;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            if (grow[rr] >= rows || skip(rr)) continue;
#pragma unroll
            for (int i = 0; i < NIE; ++i)
                if (wcol0 + 16 * i < C) *reinterpret_cast<float4*>(outp + (size_t)grow[rr] * P.ldo + colw + 16 * i) = v[rr][i];
        }
    }
    if (MODE == PG_F32 || !P.pimg) return;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (skip(rr)) continue;
        const bool rok = grow[rr] < rows;
        const float bound = s_bound[wrow0 + 16 * rr + lr];
        if (P.pbnd && rok && (nb == 0 || per_blk) && wcol0 == 0 && q == 0) P.pbnd[(size_t)nb * P.pbnd_blk_stride + grow[rr]] = bound;
        const float sc = pow2i(scale_exp(bound));
        const int rl = wrow0 + 16 * rr + lr, swz = (rl >> 2) & 3;
        char* const rowp = P.pimg + (size_t)nb * P.pimg_blk_stride +
                           (((size_t)rb * P.p_nct + P.p_kc0 + (per_blk ? 0 : nb * (C >> 4)) + (wcol0 >> 4)) * 128 + rl) * 64;
        const unsigned uhi = (unsigned)(((q >> 1) ^ swz) << 4), ulo = (unsigned)(((2 + (q >> 1)) ^ swz) << 4);
        const bool odd = q & 1;
#pragma unroll
        for (int i = 0; i < NIE; ++i) {
            if (wcol0 + 16 * i >= C) continue;
            unsigned h0, l0, h1, l1;
            split2(v[rr][i].x * sc, v[rr][i].y * sc, h0, l0);
            split2(v[rr][i].z * sc, v[rr][i].w * sc, h1, l1);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 e0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false), e1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
            const u32x4 U = {e0.x, e1.x, e0.y, e1.y};
            if (rok) __builtin_nontemporal_store(U, reinterpret_cast<u32x4*>(rowp + (size_t)i * 8192 + (odd ? ulo : uhi)));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weights: per output column c the scale 2^s_c and the L1 norm; then the image
// ---------------------------------------------------------------------------------------------------------------------
// (olen, opad): the image's row c' is row (c' / opad) olen + c' % opad of the block (zero where c' % opad >= olen): output
// columns padded per head (108 -> 112) so that the attention kernel's operands start every head at a k-chunk; olen = opad = C: identity
__device__ __forceinline__ int pg_src_row(int c, int C, int olen, int opad, int rows_src) {
    if (c >= C) return -1;
    const int r = (c / opad) * olen + c % opad;
    return (c % opad < olen && r < rows_src) ? r : -1;
}
__global__ __launch_bounds__(256) void pg_wscale_kernel(const float* __restrict__ W, int nblk, int C, int K, int BN, float* __restrict__ cinv,
                                                        float* __restrict__ wnorm, int olen, int opad, int rows_src) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= nblk * BN) return;
    const int nb = idx / BN, c = idx % BN;
    const int sr = pg_src_row(c, C, olen, opad, rows_src);
    if (sr < 0) { if (lane == 0) cinv[idx] = 1.f; return; }
    const float* wr = W + (size_t)(nb * rows_src + sr) * K;
    float mx = 0.f, l1 = 0.f;
    for (int k = lane; k < K; k += 64) { const float a = fabsf(wr[k]); mx = fmaxf(mx, a); l1 += a; }
    mx = wave_max(mx);
    l1 = wave_sum(l1);
    if (lane == 0) {
        cinv[idx] = pow2i(-scale_exp(mx));
        if (l1 == l1) atomicMax(reinterpret_cast<unsigned*>(wnorm + nb), __float_as_uint(l1 * 1.0001f));   // non-negative floats order as integers
        else wnorm[nb] = l1;
    }
}
__global__ __launch_bounds__(256) void pg_pack_kernel(const float* __restrict__ W, int nblk, int C, int K, int BN, int nct, int piece_len,
                                                      int piece_pad, const float* __restrict__ cinv, char* __restrict__ img, int olen, int opad,
                                                      int rows_src) {
    const size_t n = (size_t)nblk * nct * BN * 2, idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int half = (int)(idx & 1);
    size_t rest = idx >> 1;
    const int c = (int)(rest % BN); rest /= BN;
    const int kc = (int)(rest % nct), nb = (int)(rest / nct);
    const int sr = c < BN ? pg_src_row(c, C, olen, opad, rows_src) : -1;
    const float sc = sr >= 0 ? 1.0f / cinv[nb * BN + c] : 0.f;       // (a power of two: exact)
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kp = kc * 16 + half * 8 + e, piece = kp / piece_pad, off = kp % piece_pad, k = piece * piece_len + off;
        x[e] = (sr >= 0 && off < piece_len && k < K) ? W[(size_t)(nb * rows_src + sr) * K + k] * sc : 0.f;
    }
    uint4 hi, lo;
    split2(x[0], x[1], hi.x, lo.x); split2(x[2], x[3], hi.y, lo.y); split2(x[4], x[5], hi.z, lo.z); split2(x[6], x[7], hi.w, lo.w);
    char* d = img + (((size_t)nb * nct + kc) * BN + c) * 64;
    const int swz = (c >> 2) & 3;
    *reinterpret_cast<uint4*>(d + ((half ^ swz) << 4)) = hi;
    *reinterpret_cast<uint4*>(d + (((2 + half) ^ swz) << 4)) = lo;
}

// the wide-wave layout: physical block pb = 2 nb + half-block, row c of it = column (pb & 1) * 288 + c of logical block nb
__global__ __launch_bounds__(256) void pg_pack16w_kernel(const float* __restrict__ W, int nblk, int C, int K, int nct, int piece_len, int piece_pad,
                                                         const float* __restrict__ cinv, char* __restrict__ img, int olen, int opad, int rows_src) {
    constexpr int SB = PgGeomW::BN, LBN = PgGeomW::LBN;
    const size_t n = (size_t)nblk * 2 * nct * SB * 2, idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int half = (int)(idx & 1);
    size_t rest = idx >> 1;
    const int c = (int)(rest % SB); rest /= SB;
    const int kc = (int)(rest % nct), pb = (int)(rest / nct), nb = pb >> 1, cl = (pb & 1) * SB + c;
    const int sr = pg_src_row(cl, C, olen, opad, rows_src);
    const float sc = sr >= 0 ? 1.0f / cinv[nb * LBN + cl] : 0.f;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kp = kc * 16 + half * 8 + e, piece = kp / piece_pad, off = kp % piece_pad, k = piece * piece_len + off;
        x[e] = (sr >= 0 && off < piece_len && k < K) ? W[(size_t)(nb * rows_src + sr) * K + k] * sc : 0.f;
    }
    uint4 hi, lo;
    split2(x[0], x[1], hi.x, lo.x); split2(x[2], x[3], hi.y, lo.y); split2(x[4], x[5], hi.z, lo.z); split2(x[6], x[7], hi.w, lo.w);
    char* d = img + (((size_t)pb * nct + kc) * SB + c) * 64;
    const int swz = (c >> 2) & 3;
    *reinterpret_cast<uint4*>(d + ((half ^ swz) << 4)) = hi;
    *reinterpret_cast<uint4*>(d + (((2 + half) ^ swz) << 4)) = lo;
}

// fp32 rows -> plane image, bound = max |row| (one wave per row; K <= 1024)
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ x, int ldx, int rows, int K, char* __restrict__ img,
                                                              float* __restrict__ bnd, const float* __restrict__ bnd_in) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int ng = K >> 3, nct = K >> 4;                              // 8-wide groups
    float4 v[2][2];
    float mx = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) {
            v[g][0] = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + 8 * gi);
            v[g][1] = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + 8 * gi + 4);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[g][0].x), fabsf(v[g][0].y)), fmaxf(fabsf(v[g][0].z), fabsf(v[g][0].w))));
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[g][1].x), fabsf(v[g][1].y)), fmaxf(fabsf(v[g][1].z), fabsf(v[g][1].w))));
        }
    }
    mx = wave_max(mx);
    if (bnd_in) mx = bnd_in[row];                                     // the caller's bound (must be >= the row's maximum)
    if (lane == 0) bnd[row] = mx;
    const float sc = pow2i(scale_exp(mx));
    const int rb = row >> 7, r = row & 127, swz = (r >> 2) & 3;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) {
            uint4 hi, lo;
            split2(v[g][0].x * sc, v[g][0].y * sc, hi.x, lo.x); split2(v[g][0].z * sc, v[g][0].w * sc, hi.y, lo.y);
            split2(v[g][1].x * sc, v[g][1].y * sc, hi.z, lo.z); split2(v[g][1].z * sc, v[g][1].w * sc, hi.w, lo.w);
            char* d = img + (((size_t)rb * nct + (gi >> 1)) * 128 + r) * 64;
            *reinterpret_cast<uint4*>(d + (((gi & 1) ^ swz) << 4)) = hi;
            *reinterpret_cast<uint4*>(d + (((2 + (gi & 1)) ^ swz) << 4)) = lo;
        }
    }
}
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const char* __restrict__ img, const float* __restrict__ bnd, int rows, int K,
                                                            float* __restrict__ out, int ldo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int ng = K >> 3, nct = K >> 4, rb = row >> 7, r = row & 127, swz = (r >> 2) & 3;
    const float inv = pow2i(-scale_exp(bnd[row]));
    for (int gi = lane; gi < ng; gi += 64) {
        const char* d = img + (((size_t)rb * nct + (gi >> 1)) * 128 + r) * 64;
        const f16x8 hi = *reinterpret_cast<const f16x8*>(d + (((gi & 1) ^ swz) << 4));
        const f16x8 lo = *reinterpret_cast<const f16x8*>(d + (((2 + (gi & 1)) ^ swz) << 4));
#pragma unroll
        for (int e = 0; e < 8; ++e) out[(size_t)row * ldo + 8 * gi + e] = ((float)hi[e] + (float)lo[e]) * inv;
    }
}
__global__ __launch_bounds__(256) void ln_bound_kernel(const float* __restrict__ g, const float* __restrict__ b, int C, float* __restrict__ out) {
    __shared__ float sg[4], sb[4];
    float mg = 0.f, mb = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { mg = fmaxf(mg, fabsf(g[c])); mb = fmaxf(mb, fabsf(b[c])); }
    mg = wave_max(mg); mb = wave_max(mb);
    if ((threadIdx.x & 63) == 0) { sg[threadIdx.x >> 6] = mg; sb[threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mg = fmaxf(fmaxf(sg[0], sg[1]), fmaxf(sg[2], sg[3]));
        mb = fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]));
        out[0] = (sqrtf((float)C) * mg + mb) * 1.0001f;
    }
}

using G4 = PgGeom<4, 4>;     // column blocks of up to 256 (2D-3D: C = 256, expand = two blocks): a stage is 8 KB of A + 16 KB of W
using G4H = PgGeom<4, 4, 2>;
using G7 = PgGeom<7, 4>;     // column blocks of up to 448 (3DMatch: C = 432), 4-slot ring
using G9 = PgGeom<9, 3>;     // column blocks of up to 576 (4DMatch: C = 528), 3-slot ring (the stage is 44 KB)
using G7H = PgGeom<7, 4, 2>; // the same with 64-row workgroups (4 waves): launches that would leave CUs idle
using G9H = PgGeom<9, 3, 2>;

}  // namespace

bool pgemm_shape_ok(int C) { return C > 0 && C % 16 == 0 && C <= G9::BN; }
int pgemm_bn(int C) { return C <= G4::BN ? G4::BN : (C <= G7::BN ? G7::BN : G9::BN); }
static size_t pg_bst(int C) { return (size_t)pgemm_bn(C) * 64; }

// The 576-column geometry exists in the 64-row form only: its 128-row form (8 waves, 256 registers per wave: 144 accumulators + two fragment
// sets) spilled 198-226 registers and measured 4 % slower than the 64-row form even where it fills the chip (32 pairs of 512 x 512: 147.6
// against 141.4 ms per call).
template <int TNW, int NST, int MODE>
static int configure_mode() {
    if constexpr (TNW <= 7)
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)pgemm_kernel<TNW, NST, MODE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PgGeom<TNW, NST, 4>::SMEM));
    if constexpr (TNW == 7) {
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)pgemm_kernel<TNW, NST, MODE, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PgGeom<TNW, NST, 4>::SMEM16));
    }
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)pgemm_kernel<TNW, NST, MODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PgGeom<TNW, NST, 2>::SMEM));
    return DR_OK;
}
int pgemm_configure() {
    int rc = configure_mode<4, 4, PG_F32>();
    if (rc == DR_OK) rc = configure_mode<4, 4, PG_PLANES>();
    if (rc == DR_OK) rc = configure_mode<4, 4, PG_LN>();
    if (rc == DR_OK) rc = configure_mode<7, 4, PG_F32>();
    if (rc == DR_OK) rc = configure_mode<7, 4, PG_PLANES>();
    if (rc == DR_OK) rc = configure_mode<7, 4, PG_LN>();
    if (rc == DR_OK) rc = configure_mode<9, 3, PG_F32>();
    if (rc == DR_OK) rc = configure_mode<9, 3, PG_PLANES>();
    if (rc == DR_OK) rc = configure_mode<9, 3, PG_LN>();
    if (rc == DR_OK) {
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)pgemm16w_kernel<PG_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PgGeomW::SMEM));
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)pgemm16w_kernel<PG_PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PgGeomW::SMEM));
    }
    return rc;
}

template <int TNW, int NST, int WMN>
static void pg_launch(int mode, dim3 grid, hipStream_t st, const PgBatch& g) {
    using GG = PgGeom<TNW, NST, WMN>;
    if (mode == PG_F32) hipLaunchKernelGGL((pgemm_kernel<TNW, NST, PG_F32, WMN>), grid, dim3(GG::NTHR), GG::SMEM, st, g);
    else if (mode == PG_PLANES) hipLaunchKernelGGL((pgemm_kernel<TNW, NST, PG_PLANES, WMN>), grid, dim3(GG::NTHR), GG::SMEM, st, g);
    else hipLaunchKernelGGL((pgemm_kernel<TNW, NST, PG_LN, WMN>), grid, dim3(GG::NTHR), GG::SMEM, st, g);
}

// the wide-wave geometry (weights in the PgW::sub = 2 layout): 128-row workgroups, two sub-blocks per logical column block
static int launch_pgemm16w(const PgBatch& g, hipStream_t st) {
    int maxt = 0;
    double flops = 0;
    const int mode = g.p[0].mode;
    if (mode != PG_F32 && mode != PG_PLANES) return DR_ENOSUP;
    // KSPLIT: at most half a chip of tiles, every problem with an exchange region large enough and halves of >= 5 chunks (>= 6: the split chunk is even)
    bool ksplit = env_knob("DR_PG_KSPLIT", 1) != 0 && env_knob("DR_PG_KSPLITW", 1) != 0;
    long tiles = 0;
    for (int i = 0; i < g.n; ++i) {
        const PgProblem& p = g.p[i];
        const long tl_ = (long)((p.rows + 127) / 128) * p.nblk * 2;
        ksplit = ksplit && p.xk_buf && p.xk_flags && tl_ <= p.xk_cap && (p.A1 ? (p.nc0 >= 5 && p.nc1 >= 5) : p.nc0 >= 12);
        tiles += tl_;
    }
    ksplit = ksplit && 2 * tiles <= (long)device_cu_count();
    for (int i = 0; i < g.n; ++i) {
        const PgProblem& p = g.p[i];
        const int nst = p.nc0 + (p.A1 ? p.nc1 : 0);
        // (a virtual chunk's slot must have held real weight planes before: segments of >= 5 chunks, as in the 16x16x32 form of pgemm_kernel)
        if (p.W.sub != 2 || !pgemm16w_shape_ok(p.C) || p.rows < 1 || p.nblk < 1 || p.nc0 < 5 || (p.A1 && p.nc1 < 5) || p.mode != mode) return DR_ENOSUP;
        if (p.W.nct != nst) return DR_EINVAL;
        const int tl = ((p.rows + 127) / 128 + 7) / 8 * 8 * p.nblk * 2 * (ksplit ? 2 : 1);
        maxt = tl > maxt ? tl : maxt;
        flops += 2.0 * p.rows * p.C * p.nblk * (p.k_alg > 0 ? (double)p.k_alg : 16.0 * p.W.nct);
    }
    ProfScope ps(PK_GEMM_SPLIT, flops, st);
    PgBatch gd = g;
    gd.dbg = env_knob("DR_PG_NOEPI", 0);
    for (int i = 0; i < g.n; ++i) gd.p[i].ksplit = ksplit ? 1 : 0;
    const dim3 grid(maxt, g.n);
    if (mode == PG_F32) hipLaunchKernelGGL((pgemm16w_kernel<PG_F32>), grid, dim3(PgGeomW::NTHR), PgGeomW::SMEM, st, gd);
    else hipLaunchKernelGGL((pgemm16w_kernel<PG_PLANES>), grid, dim3(PgGeomW::NTHR), PgGeomW::SMEM, st, gd);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_pgemm(const PgBatch& g, hipStream_t st) {
    if (g.n < 1 || g.n > 3) return DR_EINVAL;
    if (g.p[0].W.sub == 2) return launch_pgemm16w(g, st);
    int maxt = 0;
    double flops = 0;
    const int bn = pgemm_bn(g.p[0].C), nst_min = bn == G9::BN ? G9::NST : G7::NST;
    // 64-row workgroups when the launch would not give every CU a 128-row one (DR_PG_HALF under dr_debug_enable_env: 0 never, 2 always)
    const int half_env = env_knob("DR_PG_HALF", 1);
    const int n_cu = device_cu_count();
    long wg128 = 0;
    for (int i = 0; i < g.n; ++i) wg128 += (long)((g.p[i].rows + 127) / 128) * g.p[i].nblk;
    // 64-row workgroups up to HALF a chip of 128-row ones: above that they would run in two rounds, slower than one round of 128-row workgroups on
    // some of the CUs (cfg5 at 8 pairs: 192 of them, 28.4 -> 27.6 ms per call; DR_PG_HALF_PCT: the threshold in percent of the CU count)
    const bool half = bn == G9::BN || half_env == 2 || (half_env == 1 && wg128 * 100 < (long)n_cu * env_knob("DR_PG_HALF_PCT", 51));
    const int bm = half ? 64 : 128;
    // KSPLIT (see the kernel): a LayerNorm launch of 64-row workgroups that fills at most half the chip, every problem one k segment of at least
    // two ring depths and with an exchange buffer: two workgroups per row block, each half the k range and half the epilogue
    bool ksplit = half && g.p[0].mode == PG_LN && env_knob("DR_PG_KSPLIT", 1) != 0;
    long wg64 = 0;
    for (int i = 0; i < g.n; ++i) {
        const PgProblem& p = g.p[i];
        const long rb64 = (p.rows + 63) / 64;
        ksplit = ksplit && p.xk_buf && p.xk_flags && p.nblk == 1 && !p.A1 && p.nc0 / 2 >= nst_min && rb64 <= p.xk_cap;
        wg64 += rb64;
    }
    ksplit = ksplit && 2 * wg64 <= (long)n_cu;
    for (int i = 0; i < g.n; ++i) {
        const PgProblem& p = g.p[i];
        if (!pgemm_shape_ok(p.C) || pgemm_bn(p.C) != bn || p.rows < 1 || p.nblk < 1 || p.nc0 < 1 || p.nc0 + (p.A1 ? p.nc1 : 0) < nst_min) return DR_ENOSUP;
        if (p.W.nct != p.nc0 + (p.A1 ? p.nc1 : 0)) return DR_EINVAL;
        const int tl = ((p.rows + bm - 1) / bm + 7) / 8 * 8 * (ksplit ? 2 : p.nblk);
        maxt = tl > maxt ? tl : maxt;
        flops += 2.0 * p.rows * p.C * p.nblk * (p.k_alg > 0 ? (double)p.k_alg : 16.0 * p.W.nct);
    }
    ProfScope ps(PK_GEMM_SPLIT, flops, st);
    const int mode = g.p[0].mode;
    for (int i = 1; i < g.n; ++i)
        if (g.p[i].mode != mode) return DR_EINVAL;           // one epilogue per launch
    const dim3 grid(maxt, g.n);
    PgBatch gd = g;
    gd.dbg = env_knob("DR_PG_NOEPI", 0);
    for (int i = 0; i < g.n; ++i) gd.p[i].ksplit = ksplit ? 1 : 0;
    if (bn == G9::BN) pg_launch<9, 3, 2>(mode, grid, st, gd);
    else if (bn == G4::BN) { if (half) pg_launch<4, 4, 2>(mode, grid, st, gd); else pg_launch<4, 4, 4>(mode, grid, st, gd); }
    else {
        // the 16x16x32 main loop (128-row workgroups): a virtual chunk's slot must have held real weight planes before (segments of >= 5 chunks)
        bool m16 = env_knob("DR_PG_M16", 1) != 0;
        for (int i = 0; i < g.n; ++i) m16 = m16 && g.p[i].nc0 >= 5 && (!g.p[i].A1 || g.p[i].nc1 >= 5);
        if (half) pg_launch<7, 4, 2>(mode, grid, st, gd);
        else if (m16) {
            using GG = PgGeom<7, 4, 4>;
            if (mode == PG_F32) hipLaunchKernelGGL((pgemm_kernel<7, 4, PG_F32, 4, true>), grid, dim3(GG::NTHR), GG::SMEM16, st, gd);
            else if (mode == PG_PLANES) hipLaunchKernelGGL((pgemm_kernel<7, 4, PG_PLANES, 4, true>), grid, dim3(GG::NTHR), GG::SMEM16, st, gd);
            else hipLaunchKernelGGL((pgemm_kernel<7, 4, PG_LN, 4, true>), grid, dim3(GG::NTHR), GG::SMEM16, st, gd);
        } else pg_launch<7, 4, 4>(mode, grid, st, gd);
    }
    DR_LAUNCH_CHECK();
    return DR_OK;
}

size_t pgemm_weight_bytes(int C, int nblk, int nct) {
    size_t b = (size_t)nblk * nct * pg_bst(C);             // image
    b += (size_t)nblk * pgemm_bn(C) * 4;                   // cinv
    b += (size_t)nblk * 4;                                 // wnorm
    return (b + 255) & ~(size_t)255;
}
void pgemm_weight_view(void* buf, int C, int nblk, int nct, PgW* v) {
    char* p = (char*)buf;
    v->img = p;
    v->cinv = reinterpret_cast<const float*>(p + (size_t)nblk * nct * pg_bst(C));
    v->wnorm = v->cinv + (size_t)nblk * pgemm_bn(C);
    v->nct = nct;
    v->sub = 0;
}

bool pgemm16w_shape_ok(int C) { return C > 0 && C % 16 == 0 && C <= PgGeomW::LBN; }
size_t pgemm16w_weight_bytes(int nblk, int nct) {
    size_t b = (size_t)nblk * 2 * nct * PgGeomW::B_ST + (size_t)nblk * PgGeomW::LBN * 4 + (size_t)nblk * 4;
    return (b + 255) & ~(size_t)255;
}
void pgemm16w_weight_view(void* buf, int nblk, int nct, PgW* v) {
    char* p = (char*)buf;
    v->img = p;
    v->cinv = reinterpret_cast<const float*>(p + (size_t)nblk * 2 * nct * PgGeomW::B_ST);
    v->wnorm = v->cinv + (size_t)nblk * PgGeomW::LBN;
    v->nct = nct;
    v->sub = 2;
}
int pgemm16w_pack_weights_block(const float* W, int C, int K, int piece_len, int piece_pad, const PgW& v, int nb, hipStream_t st, int out_len,
                                int out_pad) {
    const int olen = out_len > 0 ? out_len : C, opad = out_pad > 0 ? out_pad : C, rows_src = out_len > 0 ? C / opad * olen : C;
    if (!pgemm16w_shape_ok(C) || v.sub != 2 || piece_pad % 16 || piece_len > piece_pad || piece_len < 1) return DR_EINVAL;
    const int nct = (K + piece_len - 1) / piece_len * piece_pad / 16;
    if (nct != v.nct) return DR_EINVAL;
    constexpr int LBN = PgGeomW::LBN;
    float* cinv = (float*)v.cinv + (size_t)nb * LBN;
    float* wnorm = (float*)v.wnorm + nb;
    char* img = (char*)v.img + (size_t)nb * 2 * nct * PgGeomW::B_ST;
    DR_HIP_CHECK(hipMemsetAsync(wnorm, 0, 4, st));
    hipLaunchKernelGGL(pg_wscale_kernel, dim3((LBN + 3) / 4), dim3(256), 0, st, W, 1, C, K, LBN, cinv, wnorm, olen, opad, rows_src);
    DR_LAUNCH_CHECK();
    const size_t n = (size_t)2 * nct * PgGeomW::BN * 2;
    hipLaunchKernelGGL(pg_pack16w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, 1, C, K, nct, piece_len, piece_pad,
                       (const float*)cinv, img, olen, opad, rows_src);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int pgemm_pack_weights(const float* W, int nblk, int C, int K, int piece_len, int piece_pad, void* buf, hipStream_t st) {
    if (!pgemm_shape_ok(C) || piece_pad % 16 || piece_len > piece_pad || piece_len < 1) return DR_EINVAL;
    const int nct = (K + piece_len - 1) / piece_len * piece_pad / 16, BN = pgemm_bn(C);
    PgW v;
    pgemm_weight_view(buf, C, nblk, nct, &v);
    DR_HIP_CHECK(hipMemsetAsync((void*)v.wnorm, 0, (size_t)nblk * 4, st));
    hipLaunchKernelGGL(pg_wscale_kernel, dim3((nblk * BN + 3) / 4), dim3(256), 0, st, W, nblk, C, K, BN, (float*)v.cinv, (float*)v.wnorm, C, C, C);
    DR_LAUNCH_CHECK();
    const size_t n = (size_t)nblk * nct * BN * 2;
    hipLaunchKernelGGL(pg_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, nblk, C, K, BN, nct, piece_len, piece_pad,
                       v.cinv, (char*)v.img, C, C, C);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// one block (C rows of W [C, K]) into block `nb` of a view
int pgemm_pack_weights_block(const float* W, int C, int K, int piece_len, int piece_pad, const PgW& v, int nb, hipStream_t st, int out_len,
                             int out_pad) {
    // out_len / out_pad > 0: W has (C / out_pad) * out_len rows, spread to C image rows with zero rows in the pads
    const int olen = out_len > 0 ? out_len : C, opad = out_pad > 0 ? out_pad : C, rows_src = out_len > 0 ? C / opad * olen : C;
    if (!pgemm_shape_ok(C) || piece_pad % 16 || piece_len > piece_pad || piece_len < 1) return DR_EINVAL;
    const int nct = (K + piece_len - 1) / piece_len * piece_pad / 16;
    if (nct != v.nct) return DR_EINVAL;
    const int BN = pgemm_bn(C);
    float* cinv = (float*)v.cinv + (size_t)nb * BN;
    float* wnorm = (float*)v.wnorm + nb;
    char* img = (char*)v.img + (size_t)nb * nct * pg_bst(C);
    DR_HIP_CHECK(hipMemsetAsync(wnorm, 0, 4, st));
    hipLaunchKernelGGL(pg_wscale_kernel, dim3((BN + 3) / 4), dim3(256), 0, st, W, 1, C, K, BN, cinv, wnorm, olen, opad, rows_src);
    DR_LAUNCH_CHECK();
    const size_t n = (size_t)nct * BN * 2;
    hipLaunchKernelGGL(pg_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, 1, C, K, BN, nct, piece_len, piece_pad,
                       (const float*)cinv, img, olen, opad, rows_src);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// out[g] = max over the rows of group g of bnd[row]; groups of `grp_rows` consecutive rows (one wave per group)
__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ bnd, int ngroups, int grp_rows, float* __restrict__ out) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= ngroups) return;
    float m = 0.f;
    for (int r = lane; r < grp_rows; r += 64) m = fmaxf(m, bnd[(size_t)g * grp_rows + r]);
    m = wave_max(m);
    if (lane == 0) out[g] = m;
}
int launch_group_max(const float* bnd, int ngroups, int grp_rows, float* out, hipStream_t st) {
    if (ngroups < 1) return DR_OK;
    hipLaunchKernelGGL(group_max_kernel, dim3((ngroups + 3) / 4), dim3(256), 0, st, bnd, ngroups, grp_rows, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_planes_from_f32(const float* x, int ldx, int rows, int K, char* img, float* bnd, hipStream_t st, const float* bnd_in) {
    if (K % 16 || K > 1024 || ldx % 4 || ((uintptr_t)x & 15)) return DR_ENOSUP;
    if (rows < 1) return DR_OK;
    hipLaunchKernelGGL(planes_from_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, rows, K, img, bnd, bnd_in);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int launch_planes_to_f32(const char* img, const float* bnd, int rows, int K, float* out, int ldo, hipStream_t st) {
    if (K % 16) return DR_ENOSUP;
    if (rows < 1) return DR_OK;
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, img, bnd, rows, K, out, ldo);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
__global__ __launch_bounds__(256) void absmax_blocks_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float sm[4];
    const float* xb = x + (size_t)blockIdx.x * n;
    float m = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) m = fmaxf(m, fabsf(xb[c]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])) * 1.0001f;
}
int launch_absmax_blocks(const float* x, int nblk, int n, float* out, hipStream_t st) {
    if (nblk < 1) return DR_OK;
    hipLaunchKernelGGL(absmax_blocks_kernel, dim3(nblk), dim3(256), 0, st, x, n, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
int launch_ln_bound(const float* gamma, const float* beta, int C, float* out, hipStream_t st) {
    hipLaunchKernelGGL(ln_bound_kernel, dim3(1), dim3(256), 0, st, gamma, beta, C, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace dr

// ---------------------------------------------------------------------------------------------------------------------
// C ABI of the plane-image ops (include/diffreg_hip.h)
// ---------------------------------------------------------------------------------------------------------------------
using namespace dr;

extern "C" {

size_t dr_plane_image_bytes(int rows, int K) { return (rows > 0 && K > 0 && K % 16 == 0) ? plane_image_bytes((size_t)rows, K) : 0; }

int dr_planes_from_f32(int rows, int K, const float* x, int ldx, void* image, float* bound, void* stream) {
    if (rows < 0 || K <= 0 || !x || !image || !bound || ldx < K) return DR_EINVAL;
    return launch_planes_from_f32(x, ldx, rows, K, (char*)image, bound, (hipStream_t)stream);
}

int dr_planes_from_f32_bounded(int rows, int K, const float* x, int ldx, const float* bound_in, void* image, float* bound, void* stream) {
    if (rows < 0 || K <= 0 || !x || !image || !bound || !bound_in || ldx < K) return DR_EINVAL;
    return launch_planes_from_f32(x, ldx, rows, K, (char*)image, bound, (hipStream_t)stream, bound_in);
}

static int attention_planes_entry(int P, int Lq, int Lk, int H, int d, const void* q_image, const float* q_bound, const void* k_image,
                                  const float* k_bound, const void* v_image, const float* v_bound, const uint8_t* q_mask, const uint8_t* k_mask,
                                  void* out_image, float* out_bound, int f16_single, void* stream) {
    if (P < 1 || Lq < 1 || Lk < 1 || H < 1 || d < 4 || d % 4 || !q_image || !k_image || !v_image || !q_bound || !k_bound || !v_bound || !out_image ||
        !out_bound) return DR_EINVAL;
    const int dp = (d + 15) / 16 * 16;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.H = H; a.d = d; a.qmask = q_mask; a.kmask = k_mask;
    a.nseg = P; a.q0 = 0; a.qstride = Lq; a.Lq = Lq; a.k0 = 0; a.kstride = Lk; a.Lk = Lk;
    a.scale = 1.0f / sqrtf((float)d);
    a.pimg[0] = a.pimg[1] = (char*)out_image; a.p_split = 0x7fffffff; a.p_nct = H * dp / 16; a.p_dp = dp; a.pbnd = out_bound;
    a.qimg[0] = a.qimg[1] = (const char*)q_image; a.kimg[0] = a.kimg[1] = (const char*)k_image; a.vimg[0] = a.vimg[1] = (const char*)v_image;
    a.qbnd = q_bound; a.kgb = k_bound; a.vgb = v_bound;
    a.f16_single = f16_single;
    return launch_attention(a, (hipStream_t)stream);
}

int dr_attention_planes(int P, int Lq, int Lk, int H, int d, const void* q_image, const float* q_bound, const void* k_image,
                        const float* k_bound, const void* v_image, const float* v_bound, const uint8_t* q_mask, const uint8_t* k_mask,
                        void* out_image, float* out_bound, void* stream) {
    return attention_planes_entry(P, Lq, Lk, H, d, q_image, q_bound, k_image, k_bound, v_image, v_bound, q_mask, k_mask, out_image, out_bound, 0, stream);
}
int dr_attention_planes_f16(int P, int Lq, int Lk, int H, int d, const void* q_image, const float* q_bound, const void* k_image,
                            const float* k_bound, const void* v_image, const float* v_bound, const uint8_t* q_mask, const uint8_t* k_mask,
                            void* out_image, float* out_bound, void* stream) {
    return attention_planes_entry(P, Lq, Lk, H, d, q_image, q_bound, k_image, k_bound, v_image, v_bound, q_mask, k_mask, out_image, out_bound, 1, stream);
}

int dr_planes_to_f32(int rows, int K, const void* image, const float* bound, float* out, int ldo, void* stream) {
    if (rows < 0 || K <= 0 || !out || !image || !bound || ldo < K) return DR_EINVAL;
    return launch_planes_to_f32((const char*)image, bound, rows, K, out, ldo, (hipStream_t)stream);
}

static int plane_nct(int K, int piece_len, int piece_pad) { return (K + piece_len - 1) / piece_len * piece_pad / 16; }

size_t dr_plane_weight_bytes(int nblk, int C, int K, int piece_len, int piece_pad) {
    if (nblk < 1 || !pgemm_shape_ok(C) || K < 1 || piece_len < 1 || piece_pad < piece_len || piece_pad % 16) return 0;
    return pgemm_weight_bytes(C, nblk, plane_nct(K, piece_len, piece_pad));
}

int dr_pack_weight_planes_f32(int nblk, int C, int K, int piece_len, int piece_pad, const float* W, void* packed, void* stream) {
    if (nblk < 1 || K < 1 || !W || !packed || ((uintptr_t)packed & 15)) return DR_EINVAL;
    return pgemm_pack_weights(W, nblk, C, K, piece_len, piece_pad, packed, (hipStream_t)stream);
}

size_t dr_plane_weight_bytes_wide(int nblk, int C, int K, int piece_len, int piece_pad) {
    if (nblk < 1 || !pgemm16w_shape_ok(C) || K < 1 || piece_len < 1 || piece_pad < piece_len || piece_pad % 16) return 0;
    return pgemm16w_weight_bytes(nblk, plane_nct(K, piece_len, piece_pad));
}

int dr_pack_weight_planes_wide_f32(int nblk, int C, int K, int piece_len, int piece_pad, const float* W, void* packed, void* stream) {
    if (nblk < 1 || K < 1 || !W || !packed || ((uintptr_t)packed & 15) || !pgemm16w_shape_ok(C) || piece_len < 1 || piece_pad < piece_len || piece_pad % 16)
        return DR_EINVAL;
    PgW v;
    pgemm16w_weight_view(packed, nblk, plane_nct(K, piece_len, piece_pad), &v);
    for (int nb = 0; nb < nblk; ++nb) {
        const int rc = pgemm16w_pack_weights_block(W + (size_t)nb * C * K, C, K, piece_len, piece_pad, v, nb, (hipStream_t)stream);
        if (rc) return rc;
    }
    return DR_OK;
}

int dr_ln_bound_f32(int C, const float* gamma, const float* beta, float* out, void* stream) {
    if (C < 1 || !gamma || !beta || !out) return DR_EINVAL;
    return launch_ln_bound(gamma, beta, C, out, (hipStream_t)stream);
}

int dr_bias_max_f32(int nblk, int n, const float* bias, float* out, void* stream) {
    if (nblk < 1 || n < 1 || !bias || !out) return DR_EINVAL;
    return launch_absmax_blocks(bias, nblk, n, out, (hipStream_t)stream);
}

int dr_linear_planes_f32(const dr_planes_linear* a, void* stream) {
    if (!a || a->rows < 1 || a->nblk < 1 || !a->a0 || !a->bound0 || !a->packed || a->k0 % 16 || (a->a1 && (a->k1 % 16 || !a->bound1))) return DR_EINVAL;
    if (a->mode < 0 || a->mode > 2) return DR_EINVAL;
    PgBatch g;
    memset(&g, 0, sizeof(g));
    PgProblem& p = g.p[0];
    p.A0 = (const char*)a->a0; p.bnd0 = a->bound0; p.nc0 = a->k0 / 16;
    p.A1 = (const char*)a->a1; p.bnd1 = a->bound1; p.nc1 = a->a1 ? a->k1 / 16 : 0;
    if (a->weight_layout == DR_PL_LAYOUT_WIDE) {
        if (a->mode == PG_LN || !pgemm16w_shape_ok(a->C)) return DR_EINVAL;
        pgemm16w_weight_view((void*)a->packed, a->nblk, p.nc0 + p.nc1, &p.W);
    } else if (a->weight_layout == DR_PL_LAYOUT_BLOCK) {
        pgemm_weight_view((void*)a->packed, a->C, a->nblk, p.nc0 + p.nc1, &p.W);
    } else return DR_EINVAL;
    p.nblk = a->nblk; p.rows = a->rows; p.C = a->C; p.mode = a->mode;
    p.out = a->out; p.ldo = a->ldo; p.blk_stride = a->blk_stride;
    p.cosT = a->cos_t; p.sinT = a->sin_t; p.rot_mask = a->rot_mask; p.rot_C = a->rot_C > 0 ? a->rot_C : a->C; p.scale = a->scale;
    p.pimg = (char*)a->out_image; p.p_nct = a->out_image_k / 16; p.p_kc0 = a->out_k0 / 16; p.pbnd = a->out_bound;
    p.relu = a->relu;
    p.gamma = a->gamma; p.beta = a->beta; p.resid = a->resid; p.ldr = a->ldr; p.bnd_res = a->bound_resid; p.lnB = a->ln_bound;
    p.bias = a->bias; p.bias_max = a->bias_max; p.ln_postadd = a->ln_postadd;
    if (p.bias && p.mode == PG_PLANES && !p.bias_max) return DR_EINVAL;
    if (p.bias && (((uintptr_t)p.bias & 15) || a->C % 4)) return DR_EINVAL;
    if (p.mode == PG_PLANES && p.out && (p.ldo % 4 || p.blk_stride % 4 || ((uintptr_t)p.out & 15))) return DR_EINVAL;
    if (p.mode == PG_F32 && (!p.out || p.ldo % 4 || p.blk_stride % 4 || ((uintptr_t)p.out & 15))) return DR_EINVAL;
    if (p.mode == PG_F32 && p.rot_mask && (!p.cosT || !p.sinT || p.rot_C % 4)) return DR_EINVAL;
    if (p.mode != PG_F32 && p.pimg && (!p.pbnd || a->out_image_k % 16 || a->out_k0 % 16 || a->out_k0 + a->nblk * a->C > a->out_image_k)) return DR_EINVAL;
    if (p.mode == PG_LN && (!p.gamma || !p.beta || !p.lnB || a->nblk != 1 || (p.out && (p.ldo % 4 || ((uintptr_t)p.out & 15))) ||
                            (p.resid && (p.ldr % 4 || ((uintptr_t)p.resid & 15))))) return DR_EINVAL;
    if (p.mode != PG_F32 && !p.pimg && !p.out) return DR_EINVAL;
    if (p.mode == PG_PLANES && !p.pimg) return DR_EINVAL;
    g.n = 1;
    if (a->split_workspace) {
        // [status word, 16 bytes][flags][exchange buffer]; flags zeroed per call, one launch = epoch 1
        const int bn = a->weight_layout == DR_PL_LAYOUT_WIDE ? 576 : pgemm_bn(a->C);
        if (a->split_workspace_bytes < dr_plane_split_workspace_bytes(a->C) || ((uintptr_t)a->split_workspace & 15)) return DR_EWORKSPACE;
        char* w = (char*)a->split_workspace;
        DR_HIP_CHECK(hipMemsetAsync(w, 0, 16 + pgemm_xk_flag_bytes(), (hipStream_t)stream));
        p.xk_status = (unsigned*)w; p.xk_flags = (unsigned*)(w + 16); p.xk_buf = (float*)(w + 16 + ((pgemm_xk_flag_bytes() + 255) & ~(size_t)255));
        p.xk_epoch = 1; p.xk_cap = PG_XK_MAX_RB;
        (void)bn;
    }
    return launch_pgemm(g, (hipStream_t)stream);
}

size_t dr_plane_split_workspace_bytes(int C) {
    if (C < 16 || C > 576) return 0;
    return 16 + ((pgemm_xk_flag_bytes() + 255) & ~(size_t)255) + pgemm_xk_buf_bytes(576);     // (sized for the widest geometry: 23.6 MB)
}

int dr_plane_split_status(void* split_workspace, void* stream, int clear) {
    if (!split_workspace) return DR_EINVAL;
    return sinkhorn_call_status((unsigned*)split_workspace, (hipStream_t)stream, clear != 0);
}

}  // extern "C"
