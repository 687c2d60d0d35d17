// attention.hip -- masked multi-head softmax attention on the f32-input MFMA (exact fp32 products).
//
// Replaces  a = einsum(q,k); masked_fill; a / sqrt(d); softmax; o = einsum(a, v)
// (3D/models/transformero.py:79-85) without materialising the [L,S,H] score tensor.
// One workgroup = one (segment, head, 32-query tile); its 4 waves split the 32-key tiles
// (flash-style running max / sum per wave) and are merged through LDS at the end.
//   S^T = K Q^T   (keys on the MFMA rows, queries on the lanes: a query's scores are lane-local,
//                  16 in registers + 16 in lane^32, so the row max/sum need one cross-lane step)
//   O^T = V^T P^T (d on the MFMA rows, queries on the lanes: the softmax rescale stays lane-local
//                  and P^T is consumed straight from the score registers as the B operand)
#include <cstdlib>
#include <type_traits>
#include "kernels.h"

namespace dr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


// ---- plane-image output of one wave's 32 queries (see AttnArgs::pimg): ob[q * stride + f] holds out[q][f] ------------------
typedef _Float16 ah16x2 __attribute__((ext_vector_type(2)));
typedef float aff32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void attn_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const aff32x2 f = {x0, x1};
    const ah16x2 hh = __builtin_convertvector(f, ah16x2);
    hi = __builtin_bit_cast(unsigned, hh);
    const aff32x2 hf = __builtin_convertvector(hh, aff32x2);
    const aff32x2 r = {x0 - hf.x, x1 - hf.y};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ah16x2));
}
__device__ __forceinline__ float attn_key_bound(const AttnArgs& A, int kbase, int Lk, int lane) {
    float bm = 0.f;
    for (int k = lane; k < Lk; k += 64) bm = fmaxf(bm, A.kbnd[kbase + k]);
    return wave_max(bm) * A.vnorm[0];
}
__device__ __forceinline__ void attn_store_planes(const AttnArgs& A, const float* ob, int stride, int grow0, int nq, int head, int d,
                                                  float bound, int lane) {
    const unsigned bits = __float_as_uint(bound);
    const int e = (int)((bits >> 23) & 0xff) - 127;
    const int s = (bound > 0.f && e < 128) ? min(max(14 - e, -100), 100) : 0;
    const float sc = __uint_as_float((unsigned)(127 + s) << 23);
    const int nu = A.p_dp >> 3;
    if (head == 0 && lane < 32 && lane < nq) A.pbnd[grow0 + lane] = bound;
    for (int idx = lane; idx < 32 * nu; idx += 64) {
        const int q = idx / nu, u = idx % nu;
        if (q >= nq) continue;
        // two 16-byte reads (stride and 8 u are multiples of 4 floats; columns up to the padded head width exist in the row): eight
        // scalar reads per lane hit 16 of the 64 banks -- they were the kernel's LDS bank conflicts (2.2 M of 8.5 M LDS-active cycles)
        float x[8];
        if (8 * u + 8 <= d) {
            const float4 x03 = *reinterpret_cast<const float4*>(ob + q * stride + 8 * u), x47 = *reinterpret_cast<const float4*>(ob + q * stride + 8 * u + 4);
            x[0] = x03.x * sc; x[1] = x03.y * sc; x[2] = x03.z * sc; x[3] = x03.w * sc;
            x[4] = x47.x * sc; x[5] = x47.y * sc; x[6] = x47.z * sc; x[7] = x47.w * sc;
        } else {                                                 // the unit that straddles d (d = 108, 132: columns beyond d are not in every row buffer)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) x[e8] = (8 * u + e8 < d) ? ob[q * stride + 8 * u + e8] * sc : 0.f;
        }
        uint4 hi, lo;
        attn_split2(x[0], x[1], hi.x, lo.x); attn_split2(x[2], x[3], hi.y, lo.y);
        attn_split2(x[4], x[5], hi.z, lo.z); attn_split2(x[6], x[7], hi.w, lo.w);
        const int row = grow0 + q, side = row >= A.p_split ? 1 : 0, lrow = row - (side ? A.p_split : 0);
        const int r = lrow & 127, swz = (r >> 2) & 3, kc = head * (A.p_dp >> 4) + (u >> 1);
        char* dst = A.pimg[side] + (((size_t)(lrow >> 7) * A.p_nct + kc) * 128 + r) * 64;
        *reinterpret_cast<uint4*>(dst + (((u & 1) ^ swz) << 4)) = hi;
        *reinterpret_cast<uint4*>(dst + (((2 + (u & 1)) ^ swz) << 4)) = lo;
    }
}

// key mask of one 32-key tile as bits (bit j = key k0 + j is kept): ONE byte load per lane, issued before the tile's MFMAs, and a
// ballot after them.  (Sixteen dependent byte loads per lane inside the softmax, each behind its own branch, were the critical path
// of every kernel below: ~10 of the 20 us of a single-pair launch.)
__device__ __forceinline__ unsigned char attn_mask_byte(const AttnArgs& A, int kbase, int k0, int Lk, int l31) {
    return A.kmask ? (unsigned char)A.kmask[kbase + min(k0 + l31, Lk - 1)] : (unsigned char)1;
}
__device__ __forceinline__ unsigned attn_mask_bits(unsigned char b) { return (unsigned)__ballot(b != 0); }

template <int DG, int NDT, int NWV = 4>
struct AttnGeom {
    static constexpr int DP = DG * 8;            // padded head dim for the QK^T k-loop
    static constexpr int QS = DP + 4;            // LDS row stride of the Q / K tiles
    // the V tile uses the SAME image as the K tile (row stride QS); PV reads columns up to NDT*32-1 of a
    // row, i.e. past d into the next row: those products only land in output rows (dcol >= d) that are
    // never stored.  The per-wave O tile for the merge is [dcol][32 queries], XOR-swizzled.
    static constexpr int m1 = 32 * QS + NDT * 32;         // last row may be read NDT*32 wide
    static constexpr int m2 = m1 > NDT * 32 * 32 ? m1 : NDT * 32 * 32;
    static constexpr int WBUF = (m2 + 15) / 16 * 16;      // floats per wave buffer
    static constexpr int SMEM_FLOATS = 32 * QS + NWV * WBUF + 64 * NWV;   // NWV waves deal the key tiles (8 for the single-pair
                                                                          // launches: a wave's chain is one or two key tiles long)
};

template <int DG, int NDT, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void attention_kernel(AttnArgs A) {
    using G = AttnGeom<DG, NDT, NWV>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* wbuf = smem + 32 * G::QS + (threadIdx.x >> 6) * G::WBUF;
    float* s_ml = smem + 32 * G::QS + NWV * G::WBUF;    // [NWV][32] m, [NWV][32] l

    // ---- which segment / head / query tile ---------------------------------------------------
    int seg = blockIdx.z, qbase, kbase, Lq, Lk;
    if (seg < A.nseg) {
        qbase = A.q0 + seg * A.qstride; kbase = A.k0 + seg * A.kstride; Lq = A.Lq; Lk = A.Lk;
    } else {
        seg -= A.nseg;
        qbase = A.q0b + seg * A.qstrideb; kbase = A.k0b + seg * A.kstrideb; Lq = A.Lqb; Lk = A.Lkb;
    }
    const int qt = blockIdx.x;
    if (qt * 32 >= Lq) return;
    const int head = blockIdx.y, d = A.d;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), h = lane >> 5, l31 = lane & 31;
    const int nv4 = d >> 2;                       // float4 per row (d % 4 == 0)

    const int my_q = qt * 32 + l31;
    const bool q_valid = my_q < Lq && (!A.qmask || A.qmask[qbase + my_q]);

    f32x16 acc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (Lk + 31) / 32;
    const float sc2 = A.scale * 1.4426950408889634f;
    // tile staging: lane owns DG float4 slots of the [32][DP/4] image; slot s = lane + 64 j -> row s / (DP/4)
    float4 kreg[DG], vreg[DG];
    // slot s = lane + 64 j -> row s / (DP/4), float4 column s % (DP/4); global column offset = head*d + 4*min(c4, nv4-1).  (Recomputed
    // at each use: a table of them is 2 DG registers, which the 8-wave form -- 256 registers per wave -- does not have.)
    auto load_tile = [&](const float* base, int ld, int kt, float4 (&reg)[DG]) {
        const int lim = Lk - 1 - kt * 32;                         // last valid row of this tile
        const float* tb = base + (size_t)(kbase + kt * 32) * ld + head * d;
        int ln = lane;
        asm volatile("" : "+v"(ln));                              // (opaque: the slot arithmetic is redone here, not kept live between tiles)
#pragma unroll
        for (int j = 0; j < DG; ++j) {
            const int sl = ln + 64 * j, r = sl / (G::DP / 4), c4 = sl % (G::DP / 4);
            const unsigned off = (unsigned)(min(r, lim) * ld + 4 * min(c4, nv4 - 1));      // uniform base + 32-bit lane offset
            reg[j] = *reinterpret_cast<const float4*>(tb + off);
        }
    };
    // zeroing of padding rows / columns happens at store time (touching the data earlier would wait for the load)
    auto store_tile_k = [&](const float4 (&reg)[DG], int kt) {
        const int lim = Lk - 1 - kt * 32;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < DG; ++j) {
            const int sl = ln + 64 * j, r = sl / (G::DP / 4), c4 = sl % (G::DP / 4);
            const bool ok = r <= lim && c4 < nv4;
            const float4 v = reg[j];
            *reinterpret_cast<float4*>(wbuf + r * G::QS + 4 * c4) =
                make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };
    if (w < nkt) load_tile(A.k, A.ldk, w, kreg);
    // ---- stage the Q tile (zero padded): every load of it is in flight together, behind the first K tile's ---------------------
    {
        constexpr int NQ = (32 * (G::DP / 4) + 64 * NWV - 1) / (64 * NWV);
        float4 qv[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int sq = t + 64 * NWV * i, r = sq / (G::DP / 4), c4 = sq % (G::DP / 4), qr = qt * 32 + r;
            qv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sq < 32 * (G::DP / 4) && qr < Lq && c4 < nv4)
                qv[i] = *reinterpret_cast<const float4*>(A.q + (size_t)(qbase + qr) * A.ldq + head * d + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int sq = t + 64 * NWV * i, r = sq / (G::DP / 4), c4 = sq % (G::DP / 4);
            if (sq < 32 * (G::DP / 4)) *reinterpret_cast<float4*>(Qs + r * G::QS + 4 * c4) = qv[i];
        }
    }
    __syncthreads();
    for (int kt = w; kt < nkt; kt += NWV) {
        // ---- K tile -> LDS; V tile of the same keys starts loading -------------------------------------
        const unsigned char mbyte = attn_mask_byte(A, kbase, kt * 32, Lk, l31);
        wave_lds_fence();
        store_tile_k(kreg, kt);
        __builtin_amdgcn_sched_barrier(0);                       // (the V loads after the K stores: the two staging sets never live together)
        load_tile(A.v, A.ldv, kt, vreg);
        wave_lds_fence();
        // ---- S^T = K Q^T ----------------------------------------------------------------------------
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const float* kp = wbuf + l31 * G::QS + 4 * h;
        const float* qp = Qs + l31 * G::QS + 4 * h;
        constexpr int UG = (NWV == 8 && DG % 2 == 0) ? DG / 2 : DG;   // (8 waves = 256 registers per wave: bound the fragment preloading)
#pragma unroll UG
        for (int g = 0; g < DG; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(kp + 8 * g);
            const float4 b = *reinterpret_cast<const float4*>(qp + 8 * g);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, sc, 0, 0, 0);
        }
        // ---- mask, scale, running softmax (register r holds key (r&3) + 8(r>>2) + 4h of the tile) ----
        const unsigned kbits = attn_mask_bits(mbyte) >> (4 * h);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float s = sc[r];
            // transformero.py:82: fill where the query is valid and the key is not; keys beyond the
            // segment exist only as tile padding
            const bool drop = kk >= Lk || (q_valid && !((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u));     // transformero.py:82
            s = drop ? -INFINITY : s * sc2;                     // log2-domain logits: 2^(s*scale*log2 e)
            sc[r] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        float alpha = 1.f, psum = 0.f;
        if (m_new == -INFINITY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        } else {
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc[r] - m_new);
                sc[r] = p;
                psum += p;
            }
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        // ---- V tile -> LDS (same buffer); the K tile of this wave's next keys starts loading -------------
        wave_lds_fence();
        store_tile_k(vreg, kt);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NWV < nkt) load_tile(A.k, A.ldk, kt + NWV, kreg);
        wave_lds_fence();
        // ---- O^T += V^T P^T : step r contracts keys (r&3)+8(r>>2) (h = 0 lanes) and +4 (h = 1 lanes) ----
        // MFMA row l31 of tile i is feature NDT*l31 + i (any bijection works: output rows are only labels), so the
        // NDT operands of a step are NDT consecutive floats of one V row: one vector LDS read per step
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
          if (NWV == 8) __builtin_amdgcn_sched_barrier(0);          // (bounds the fragment preloading of the 8-wave form)
#pragma unroll
          for (int r = r0; r < r0 + 4; ++r) {
            const float* vp = wbuf + ((r & 3) + 8 * (r >> 2) + 4 * h) * G::QS + NDT * l31;
            float vv[NDT];
            if (NDT == 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(vp);
                vv[0] = t4.x; vv[1] = t4.y; vv[2] = t4.z; vv[3] = t4.w;
            } else if (NDT == 2) {
                const float2 t2 = *reinterpret_cast<const float2*>(vp);
                vv[0] = t2.x; vv[1] = t2.y;
            } else {
#pragma unroll
                for (int i = 0; i < NDT; ++i) vv[i] = vp[i];
            }
#pragma unroll
            for (int i = 0; i < NDT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[i], sc[r], acc[i], 0, 0, 0);
          }
        }
    }

    // ---- merge the NWV waves: out = sum_w e^{m_w - m*} O_w / sum_w e^{m_w - m*} l_w ---------------------
    // per-wave O tile image: [dcol][q ^ (dcol & 31)] (conflict-free for the lane = q writes and the lane = dcol reads)
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dc = NDT * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
            wbuf[dc * 32 + (l31 ^ (dc & 31))] = acc[i][r];
        }
    if (h == 0) {
        s_ml[w * 32 + l31] = m_run;
        s_ml[NWV * 32 + w * 32 + l31] = l_run;
    }
    __syncthreads();
    const float* W0 = smem + 32 * G::QS;
    for (int idx = t; idx < 32 * d; idx += 64 * NWV) {
        const int q = idx / d, c = idx % d;
        if (qt * 32 + q >= Lq) continue;
        float ms = -INFINITY;
#pragma unroll
        for (int k = 0; k < NWV; ++k) ms = fmaxf(ms, s_ml[k * 32 + q]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            const float mk = s_ml[k * 32 + q];
            const float e = (mk == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mk - ms);
            num = fmaf(e, W0[k * G::WBUF + c * 32 + (q ^ (c & 31))], num);
            den = fmaf(e, s_ml[NWV * 32 + k * 32 + q], den);
        }
        A.out[(size_t)(qbase + qt * 32 + q) * A.ldo + head * d + c] = num / den;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Flash form: one workgroup = one (segment, head, 128-query block); wave w owns queries 32 w .. 32 w + 31 of the
// block for the whole key range (no merge), its Q fragments live in registers, and the 4 waves share every K / V
// tile, which all 256 threads stage together (global -> registers one tile ahead -> the other LDS buffer).
// Against the kernel above (4 waves x their own K/V tiles of a 32-query block): a K/V tile is fetched and written
// to LDS once per 128 queries instead of once per 32, there is no cross-wave merge, and a wave runs 8 key tiles
// back to back instead of 2 between a prologue and an epilogue.  Same arithmetic per (query, key).
template <int DG, int NDT>
struct FlashGeom {
    static constexpr int DP = DG * 8, QS = DP + 4;
    static constexpr int TILE = 32 * QS;                         // floats per K or V tile image
    static constexpr int SLOTS = (32 * (DP / 4) + 255) / 256;    // float4 staging slots per thread and tile
    static constexpr int SMEM_FLOATS = 4 * TILE + NDT * 32 + 64; // 2 buffers x (K | V); PV over-reads past the last row
};

template <int DG, int NDT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_flash_kernel(AttnArgs A) {
    using G = FlashGeom<DG, NDT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    int seg = blockIdx.z, qbase, kbase, Lq, Lk;
    if (seg < A.nseg) {
        qbase = A.q0 + seg * A.qstride; kbase = A.k0 + seg * A.kstride; Lq = A.Lq; Lk = A.Lk;
    } else {
        seg -= A.nseg;
        qbase = A.q0b + seg * A.qstrideb; kbase = A.k0b + seg * A.kstrideb; Lq = A.Lqb; Lk = A.Lkb;
    }
    const int qb = blockIdx.x * 128;
    if (qb >= Lq) return;
    const int head = blockIdx.y, d = A.d;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, l31 = lane & 31;
    const int nv4 = d >> 2;

    // ---- Q fragments of this lane's query: group g holds k = 8 g + 4 h .. + 3 (zero past d / past Lq) ---------
    const int my_q = qb + w * 32 + l31;
    const bool q_in = my_q < Lq;
    const bool q_valid = q_in && (!A.qmask || A.qmask[qbase + my_q]);
    float4 qf[DG];
    {
        const float* qrow = A.q + (size_t)(qbase + min(my_q, Lq - 1)) * A.ldq + head * d;
#pragma unroll
        for (int g = 0; g < DG; ++g) {
            const int c4 = 2 * g + h;
            float4 v = *reinterpret_cast<const float4*>(qrow + 4 * min(c4, nv4 - 1));
            if (!(q_in && c4 < nv4)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            qf[g] = v;
        }
    }

    // ---- K / V staging slots --------------------------------------------------------------------------------
    int srow[G::SLOTS], slds[G::SLOTS], scol[G::SLOTS];
    bool scv[G::SLOTS], sact[G::SLOTS];
#pragma unroll
    for (int j = 0; j < G::SLOTS; ++j) {
        const int sl = t + 256 * j, slc = min(sl, 32 * (G::DP / 4) - 1);
        const int r = slc / (G::DP / 4), c4 = slc % (G::DP / 4);
        sact[j] = sl < 32 * (G::DP / 4);
        srow[j] = r; slds[j] = r * G::QS + 4 * c4; scv[j] = c4 < nv4;
        scol[j] = head * d + 4 * min(c4, nv4 - 1);
    }
    float4 kreg[G::SLOTS], vreg[G::SLOTS];
    auto load_tiles = [&](int kt) {
        const int lim = Lk - 1 - kt * 32;
        const float* kb = A.k + (size_t)(kbase + kt * 32) * A.ldk;
        const float* vb = A.v + (size_t)(kbase + kt * 32) * A.ldv;
#pragma unroll
        for (int j = 0; j < G::SLOTS; ++j) {
            const int r = min(srow[j], lim);
            kreg[j] = *reinterpret_cast<const float4*>(kb + (size_t)r * A.ldk + scol[j]);
            vreg[j] = *reinterpret_cast<const float4*>(vb + (size_t)r * A.ldv + scol[j]);
        }
    };
    auto store_tiles = [&](int kt) {
        const int lim = Lk - 1 - kt * 32;
        float* Kt = smem + (kt & 1) * 2 * G::TILE;
        float* Vt = Kt + G::TILE;
#pragma unroll
        for (int j = 0; j < G::SLOTS; ++j) {
            if (!sact[j]) continue;
            const bool ok = srow[j] <= lim && scv[j];
            const float4 kv = kreg[j], vv = vreg[j];
            *reinterpret_cast<float4*>(Kt + slds[j]) = make_float4(ok ? kv.x : 0.f, ok ? kv.y : 0.f, ok ? kv.z : 0.f, ok ? kv.w : 0.f);
            *reinterpret_cast<float4*>(Vt + slds[j]) = make_float4(ok ? vv.x : 0.f, ok ? vv.y : 0.f, ok ? vv.z : 0.f, ok ? vv.w : 0.f);
        }
    };

    f32x16 acc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int nkt = (Lk + 31) / 32;
    const float sc2 = A.scale * 1.4426950408889634f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tiles(kt + 1);
        const unsigned char mbyte = attn_mask_byte(A, kbase, kt * 32, Lk, l31);
        const float* Kt = smem + (kt & 1) * 2 * G::TILE;
        const float* Vt = Kt + G::TILE;
        // ---- S^T = K Q^T ----------------------------------------------------------------------------
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const float* kp = Kt + l31 * G::QS + 4 * h;
#pragma unroll
        for (int g = 0; g < DG; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(kp + 8 * g);
            const float4 b = qf[g];
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, sc, 0, 0, 0);
        }
        // ---- mask, scale, running softmax (register r holds key (r&3) + 8(r>>2) + 4h of the tile) ----
        const unsigned kbits = attn_mask_bits(mbyte) >> (4 * h);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float s = sc[r];
            const bool drop = kk >= Lk || (q_valid && !((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u));     // transformero.py:82
            s = drop ? -INFINITY : s * sc2;
            sc[r] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        float alpha = 1.f, psum = 0.f;
        if (m_new == -INFINITY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        } else {
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc[r] - m_new);
                sc[r] = p;
                psum += p;
            }
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        // ---- O^T += V^T P^T (MFMA row l31 of tile i is feature NDT * l31 + i) ---------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vp = Vt + ((r & 3) + 8 * (r >> 2) + 4 * h) * G::QS + NDT * l31;
            float vv[NDT];
            if (NDT == 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(vp);
                vv[0] = t4.x; vv[1] = t4.y; vv[2] = t4.z; vv[3] = t4.w;
            } else if (NDT == 2) {
                const float2 t2 = *reinterpret_cast<const float2*>(vp);
                vv[0] = t2.x; vv[1] = t2.y;
            } else {
#pragma unroll
                for (int i = 0; i < NDT; ++i) vv[i] = vp[i];
            }
#pragma unroll
            for (int i = 0; i < NDT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[i], sc[r], acc[i], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tiles(kt + 1);                   // the other buffer: last read before the previous barrier
        __syncthreads();
    }

    // ---- out[q][f] = O^T[f][q] / l : through LDS (the tile buffers are free) so that rows are written contiguously
    float* ob = smem + w * G::TILE;                              // [32 queries][QS]
    const float inv = 1.0f / l_run;                              // a fully masked query gives 0/0 = NaN like the reference softmax
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = NDT * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
            if (f < G::QS) ob[l31 * G::QS + f] = acc[i][r] * inv;
        }
    wave_lds_fence();
    if (A.pimg[0]) {
        attn_store_planes(A, ob, G::QS, qbase + qb + w * 32, Lq - (qb + w * 32), head, d, attn_key_bound(A, kbase, Lk, lane), lane);
        return;
    }
    for (int idx = lane; idx < 32 * nv4; idx += 64) {
        const int q = idx / nv4, c4 = idx % nv4;
        const int qq = qb + w * 32 + q;
        if (qq < Lq)
            *reinterpret_cast<float4*>(A.out + (size_t)(qbase + qq) * A.ldo + head * d + 4 * c4) =
                *reinterpret_cast<const float4*>(ob + q * G::QS + 4 * c4);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Flash form on the bf16 pipe: the same workgroup / wave decomposition, but every fp32 product of QK^T and PV is
// computed as in the wide GEMM (gemm.hip): operands written as hi + mid + lo bf16 (exact to 2^-27), the six
// significant bf16 x bf16 products accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest first -- fp32-level
// accuracy at 2.67x the f32-input MFMA rate.
//   * K tile  -> LDS as [32 keys][3 planes][DPS k] bf16 (A operand of S^T = K Q^T: a lane reads 8 consecutive k)
//   * V tile  -> LDS TRANSPOSED as [feature][3 planes][32 keys] bf16, keys in the order in which the score
//                registers of a lane hold them (A operand of O^T = V^T P^T: a lane reads 8 consecutive keys of its
//                feature); a thread stages a 4-key x 4-feature block, so the transposition is four 8-byte stores
//   * Q stays in fp32 registers and is split k-step by k-step between the MFMAs; P is split from the score registers
//   * K(kt+1) is written after the barrier that ends QK^T(kt), V(kt+1) after the one that ends PV(kt): two barriers
//     per key tile, single K and V images (49 KB), staging registers for one of them at a time.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef float af32x2 __attribute__((ext_vector_type(2)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void asplit_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const af32x2 f = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, abf16x2));
    const af32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, abf16x2));
    const af32x2 r2 = {r.x - __uint_as_float(mid << 16), r.y - __uint_as_float(mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, abf16x2));
}

template <int KS, int NDT>
struct FlashSplitGeom {
    static constexpr int DPS = 16 * KS;                          // padded head dim (k-steps of 16)
    static constexpr int KROW = 6 * DPS + 16;                    // bytes per key row of the K image (3 planes + pad)
    static constexpr int KIMG = 32 * KROW;
    static constexpr int VROW = 3 * 64 + 16;                     // bytes per feature row of the V^T image
    static constexpr int VIMG = NDT * 32 * VROW;
    static constexpr int OQS = NDT * 32 + 4;                     // floats per query row of the output transposition
    static constexpr int OBYTES = 4 * 32 * OQS * 4;
    static constexpr int SMEM = (KIMG + VIMG > OBYTES ? KIMG + VIMG : OBYTES) + 64;
    static constexpr int KSLOTS = (32 * (DPS / 4) + 255) / 256;  // float4 staging slots per thread (K)
    static constexpr int VGROUPS = 8 * (DPS / 4);                // 4-key x 4-feature blocks of a V tile (<= 256)
};

template <int KS, int NDT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_flash_split_kernel(AttnArgs A) {
    using G = FlashSplitGeom<KS, NDT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    char* const Kimg = lds;
    char* const Vimg = lds + G::KIMG;

    int seg = blockIdx.z, qbase, kbase, Lq, Lk;
    if (seg < A.nseg) {
        qbase = A.q0 + seg * A.qstride; kbase = A.k0 + seg * A.kstride; Lq = A.Lq; Lk = A.Lk;
    } else {
        seg -= A.nseg;
        qbase = A.q0b + seg * A.qstrideb; kbase = A.k0b + seg * A.kstrideb; Lq = A.Lqb; Lk = A.Lkb;
    }
    const int qb = blockIdx.x * 128;
    if (qb >= Lq) return;
    const int head = blockIdx.y, d = A.d;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, h = lane >> 5, l31 = lane & 31;
    const int nv4 = d >> 2;

    // ---- Q of this lane's query in fp32: k-step s needs k = 16 s + 8 h .. + 7 (two float4), zero past d / past Lq
    const int my_q = qb + w * 32 + l31;
    const bool q_in = my_q < Lq;
    const bool q_valid = q_in && (!A.qmask || A.qmask[qbase + my_q]);
    float4 qf[KS][2];
    {
        const float* qrow = A.q + (size_t)(qbase + min(my_q, Lq - 1)) * A.ldq + head * d;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c4 = 4 * s + 2 * h + e;
                float4 v = *reinterpret_cast<const float4*>(qrow + 4 * min(c4, nv4 - 1));
                if (!(q_in && c4 < nv4)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                qf[s][e] = v;
            }
    }

    // ---- staging maps ------------------------------------------------------------------------------------------
    // K: slot = (key row r, float4 column c4) -> planes at Kimg + r KROW + plane 2 DPS + 8 c4 (4 bf16 = 8 bytes)
    int krow[G::KSLOTS], kcol[G::KSLOTS], klds[G::KSLOTS];
    bool kcv[G::KSLOTS], kact[G::KSLOTS];
#pragma unroll
    for (int j = 0; j < G::KSLOTS; ++j) {
        const int sl = t + 256 * j, slc = min(sl, 32 * (G::DPS / 4) - 1);
        const int r = slc / (G::DPS / 4), c4 = slc % (G::DPS / 4);
        kact[j] = sl < 32 * (G::DPS / 4);
        krow[j] = r; klds[j] = r * G::KROW + 8 * c4; kcv[j] = c4 < nv4;
        kcol[j] = head * d + 4 * min(c4, nv4 - 1);
    }
    // V: block = (key quad kq = 0..7, feature quad c4): keys 4 kq .. 4 kq + 3 land at position
    //    16 (kq >> 2) + 8 (kq & 1) + 4 ((kq >> 1) & 1) of rows 4 c4 .. 4 c4 + 3 (the order of a lane's score registers)
    const bool vact = t < G::VGROUPS;
    const int vkq = min(t, G::VGROUPS - 1) / (G::DPS / 4), vc4 = min(t, G::VGROUPS - 1) % (G::DPS / 4);
    const bool vcv = vc4 < nv4;
    const int vcol = head * d + 4 * min(vc4, nv4 - 1);
    const int vlds = 4 * vc4 * G::VROW + 2 * (16 * (vkq >> 2) + 8 * (vkq & 1) + 4 * ((vkq >> 1) & 1));
    float4 kreg[G::KSLOTS], vreg[4];
    auto load_k = [&](int kt) {
        const int lim = Lk - 1 - kt * 32;
        const float* kb = A.k + (size_t)(kbase + kt * 32) * A.ldk;
#pragma unroll
        for (int j = 0; j < G::KSLOTS; ++j) kreg[j] = *reinterpret_cast<const float4*>(kb + (size_t)min(krow[j], lim) * A.ldk + kcol[j]);
    };
    auto load_v = [&](int kt) {
        const int lim = Lk - 1 - kt * 32;
        const float* vb = A.v + (size_t)(kbase + kt * 32) * A.ldv;
#pragma unroll
        for (int e = 0; e < 4; ++e) vreg[e] = *reinterpret_cast<const float4*>(vb + (size_t)min(4 * vkq + e, lim) * A.ldv + vcol);
    };
    auto store_k = [&](int kt) {
        const int lim = Lk - 1 - kt * 32;
#pragma unroll
        for (int j = 0; j < G::KSLOTS; ++j) {
            if (!kact[j]) continue;
            const bool ok = krow[j] <= lim && kcv[j];
            const float4 v = kreg[j];
            uint2 hi, mid, lo;
            asplit_pair(ok ? v.x : 0.f, ok ? v.y : 0.f, hi.x, mid.x, lo.x);
            asplit_pair(ok ? v.z : 0.f, ok ? v.w : 0.f, hi.y, mid.y, lo.y);
            char* dst = Kimg + klds[j];
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + 2 * G::DPS) = mid;
            *reinterpret_cast<uint2*>(dst + 4 * G::DPS) = lo;
        }
    };
    auto store_v = [&](int kt) {
        if (!vact) return;
        const int lim = Lk - 1 - kt * 32;
        float x[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = 4 * vkq + e <= lim && vcv;
            x[e][0] = ok ? vreg[e].x : 0.f; x[e][1] = ok ? vreg[e].y : 0.f; x[e][2] = ok ? vreg[e].z : 0.f; x[e][3] = ok ? vreg[e].w : 0.f;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            uint2 hi, mid, lo;
            asplit_pair(x[0][f], x[1][f], hi.x, mid.x, lo.x);
            asplit_pair(x[2][f], x[3][f], hi.y, mid.y, lo.y);
            char* dst = Vimg + vlds + f * G::VROW;
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + 64) = mid;
            *reinterpret_cast<uint2*>(dst + 128) = lo;
        }
    };

    f32x16 acc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int nkt = (Lk + 31) / 32;
    const float sc2 = A.scale * 1.4426950408889634f;

    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_k(kt + 1);
        const unsigned char mbyte = attn_mask_byte(A, kbase, kt * 32, Lk, l31);
        // ---- S^T = K Q^T ----------------------------------------------------------------------------------
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const char* kp = Kimg + l31 * G::KROW + 16 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 qh, qm, ql;
            asplit_pair(qf[s][0].x, qf[s][0].y, qh.x, qm.x, ql.x);
            asplit_pair(qf[s][0].z, qf[s][0].w, qh.y, qm.y, ql.y);
            asplit_pair(qf[s][1].x, qf[s][1].y, qh.z, qm.z, ql.z);
            asplit_pair(qf[s][1].z, qf[s][1].w, qh.w, qm.w, ql.w);
            const abf16x8 k0 = *reinterpret_cast<const abf16x8*>(kp + 32 * s);
            const abf16x8 k1 = *reinterpret_cast<const abf16x8*>(kp + 2 * G::DPS + 32 * s);
            const abf16x8 k2 = *reinterpret_cast<const abf16x8*>(kp + 4 * G::DPS + 32 * s);
            const au32x4 qhv = {qh.x, qh.y, qh.z, qh.w}, qmv = {qm.x, qm.y, qm.z, qm.w}, qlv = {ql.x, ql.y, ql.z, ql.w};
            const abf16x8 q0 = __builtin_bit_cast(abf16x8, qhv), q1 = __builtin_bit_cast(abf16x8, qmv), q2 = __builtin_bit_cast(abf16x8, qlv);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, q0, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q2, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, sc, 0, 0, 0);
        }
        __syncthreads();                                         // every wave is done with the K image
        if (more) {
            store_k(kt + 1);
            load_v(kt + 1);
        }
        // ---- mask, scale, running softmax (register r holds key (r&3) + 8(r>>2) + 4h of the tile) ----------
        const unsigned kbits = attn_mask_bits(mbyte) >> (4 * h);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float s = sc[r];
            const bool drop = kk >= Lk || (q_valid && !((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u));     // transformero.py:82
            s = drop ? -INFINITY : s * sc2;
            sc[r] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        float alpha = 1.f, psum = 0.f;
        if (m_new == -INFINITY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        } else {
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sc[r] - m_new);
                sc[r] = p;
                psum += p;
            }
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        // ---- O^T += V^T P^T : k-step s contracts the keys of score registers 8 s .. 8 s + 7 -------------------
        const char* vp = Vimg + l31 * G::VROW + 16 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 ph, pm, pl;
            asplit_pair(sc[8 * s + 0], sc[8 * s + 1], ph.x, pm.x, pl.x);
            asplit_pair(sc[8 * s + 2], sc[8 * s + 3], ph.y, pm.y, pl.y);
            asplit_pair(sc[8 * s + 4], sc[8 * s + 5], ph.z, pm.z, pl.z);
            asplit_pair(sc[8 * s + 6], sc[8 * s + 7], ph.w, pm.w, pl.w);
            const au32x4 phv = {ph.x, ph.y, ph.z, ph.w}, pmv = {pm.x, pm.y, pm.z, pm.w}, plv = {pl.x, pl.y, pl.z, pl.w};
            const abf16x8 p0 = __builtin_bit_cast(abf16x8, phv), p1 = __builtin_bit_cast(abf16x8, pmv), p2 = __builtin_bit_cast(abf16x8, plv);
#pragma unroll
            for (int i = 0; i < NDT; ++i) {
                const char* vr = vp + i * 32 * G::VROW + 32 * s;
                const abf16x8 v0 = *reinterpret_cast<const abf16x8*>(vr);
                const abf16x8 v1 = *reinterpret_cast<const abf16x8*>(vr + 64);
                const abf16x8 v2 = *reinterpret_cast<const abf16x8*>(vr + 128);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, p0, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p2, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p0, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p1, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p0, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();                                         // every wave is done with the V image
        if (more) store_v(kt + 1);
        // (the K image written above is read after this barrier, the V image after the next one)
    }
    __syncthreads();

    // ---- out[q][f] = O^T[f][q] / l, through LDS; MFMA row l31 of tile i is feature 32 i + l31 ---------------------
    float* ob = smem + w * 32 * G::OQS;
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[l31 * G::OQS + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[i][r] * inv;
    wave_lds_fence();
    if (A.pimg[0]) {
        attn_store_planes(A, ob, G::OQS, qbase + qb + w * 32, Lq - (qb + w * 32), head, d, attn_key_bound(A, kbase, Lk, lane), lane);
        return;
    }
    for (int idx = lane; idx < 32 * nv4; idx += 64) {
        const int q = idx / nv4, c4 = idx % nv4;
        const int qq = qb + w * 32 + q;
        if (qq < Lq)
            *reinterpret_cast<float4*>(A.out + (size_t)(qbase + qq) * A.ldo + head * d + 4 * c4) =
                *reinterpret_cast<const float4*>(ob + q * G::OQS + 4 * c4);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Flash form on fp16 hi / lo plane images (the plane path of the loop, pgemm.h): Q, K, V arrive as the images the q|k|v GEMM's
// epilogue wrote (rotary applied, heads padded to p_dp = 16 KS features, one power-of-two scale per query row / per key group),
// so nothing is split on the VALU except P, and every fp32 product of Q K^T and P V is THREE fp16 MFMAs (hi*hi + hi*lo + lo*hi)
// instead of the six bf16 ones of attention_flash_split_kernel:
//   * a K tile (32 keys x KS chunks x 64 B) and a V tile land in LDS by LDS-DMA -- 28 one-KB instructions per tile, issued one tile
//     ahead into the other buffer, one barrier per tile; the image's own layout IS the LDS layout;
//   * S^T = K Q^T: K fragments by ds_read_b128 (the image's unit swizzle makes them conflict-free), Q fragments (both planes of
//     all KS chunks: 8 KS registers) are loaded once per wave straight from the image;
//   * O^T = V^T P^T: V is stored key-major, its A fragments (feature rows, 8 keys each) come out of ds_read_b64_tr_b16, two 4-key
//     blocks per fragment in the order in which a lane's score registers hold the keys, so P goes from registers into the MFMA;
//   * the scales: S = (K' Q'^T) 2^-(sk + sq) in fp32 behind the MFMA, O = (V'^T P^T) 2^-sv at the end (exact powers of two).
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 ah16x8 __attribute__((ext_vector_type(8)));
typedef short as16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void a_lds_void;
typedef __attribute__((address_space(3))) as16x4 a_lds_s4;

// op(own value, the value of the lane 32 away): one v_permlane32_swap instead of a ds_bpermute round trip through the LDS crossbar
template <typename Op>
__device__ __forceinline__ float attn_xhalf(float v, Op op) {
    typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(v);
    const au32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);       // (low low), (high high): both halves see both values
    return op(__uint_as_float(r.x), __uint_as_float(r.y));
}

template <int KS, int NDT>
struct AttnPlGeom {
    static constexpr int KCH = 2048;                             // a K chunk: 32 keys x 64 B
    static constexpr int VCH = 2048 + 32;                        // a V chunk, shifted by 32 B per chunk: the two 16-lane groups of a
                                                                 // transposed read (features 0..15 / 16..31) then hit disjoint banks
    static constexpr int KIMG = KS * KCH, VIMG = KS * VCH, BUF = KIMG + VIMG;
    // NB tile buffers (the DMA of tile kt + NB - 1 is issued when tile kt starts).  Two.  Round 5 measured four at d = 64 (66 KB, counted waits
    // that leave the two younger tiles' pieces in flight): 141 against 137 us for 8 x 2048 x 2048 -- the prefetch distance is not what bounds the
    // kernel (profiles/r05_attention_planes_d64_experiments.json, experiment 4); the ring form stays in the loop below for NB > 2.
    static constexpr int NB = 2;
    static constexpr int OQS = NDT * 32 + 4;
    static constexpr int OBYTES = 4 * 32 * OQS * 4;
    static constexpr int SMEM = (NB * BUF > OBYTES ? NB * BUF : OBYTES) + 64;
    // KG = 2 (two key groups of four waves, see the kernel): every group its own tile buffers; the merge region [4 waves][NDT 16 + 2][64 lanes] floats
    static constexpr int MBYTES = 4 * (NDT * 16 + 2) * 64 * 4;
    static constexpr int SMEM2_ = 2 * NB * BUF > OBYTES ? 2 * NB * BUF : OBYTES;
    static constexpr int SMEM2 = (SMEM2_ > MBYTES ? SMEM2_ : MBYTES) + 64;
};

__device__ __forceinline__ int attn_scale_exp(float bound) {
    const unsigned bits = __float_as_uint(bound);
    const int e = (int)((bits >> 23) & 0xff) - 127;
    return (bound > 0.f && e < 128) ? min(max(14 - e, -100), 100) : 0;
}

// (d = 64 uses 141 registers: three workgroups per CU.  Forced into 128 -- four per CU -- it spills 13 registers and measured 1-5 % slower
//  at every cfg5 shape: profiles/r04_attention_planes_softmax_trims.json)
// F16 (opt-in, DR_LOOP_ATTN_F16): ONE fp16 product per contraction -- the hi planes of q, k, v and of P only -- instead of the three that make an
// fp32-grade product: what BASELINE's cfg3 / cfg5 wording ("bf16 / fp16 MFMA attention") literally asks for.  11-bit operands: not the default.
// KG = 2 (round 6; launches of at most one workgroup per CU -- cfg3: 16 segments x 4 heads x 4 query blocks = 256 -- ran ONE wave per SIMD,
// MFMA busy 0.12): EIGHT waves, two key groups of four; group g takes the key tiles g, g + 2, .. of the same 128 queries with its own tile
// buffers and its own running softmax; the groups are merged through LDS at the end (m = max, rescale, add) by group 0, which writes the output.
template <int KS, int NDT, bool F16 = false, int KG = 1>
__global__ __launch_bounds__(256 * KG) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_planes_kernel(AttnArgs A) {
    using G = AttnPlGeom<KS, NDT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds_base = (unsigned)(size_t)(a_lds_void*)lds;

    // workgroup -> (query block, head, segment).  Hardware order: x fastest, ids dealt round-robin to the 8 XCDs -- the query blocks of one
    // (head, segment), which all stream the SAME K / V tiles, would land on all 8 XCDs and every XCD's L2 would fetch those tiles from memory
    // (PMC at cfg5's image self-attention: 302 MB per launch = q + o + EIGHT times k | v).  With xcd_groups a chunk of 8 gridDim.x consecutive ids
    // is dealt so that XCD j takes every query block of the chunk's j-th (head, segment): K / V cross the fabric once.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.xcd_groups) {
        const int NX = gridDim.x, NY = gridDim.y;
        const int lin = bx + NX * (by + NY * bz), chunk = lin / (8 * NX), r = lin % (8 * NX), grp = chunk * 8 + (r & 7);
        bx = r >> 3; by = grp % NY; bz = grp / NY;
    }
    int seg = bz, qbase, kbase, Lq, Lk;
    if (seg < A.nseg) {
        qbase = A.q0 + seg * A.qstride; kbase = A.k0 + seg * A.kstride; Lq = A.Lq; Lk = A.Lk;
    } else {
        seg -= A.nseg;
        qbase = A.q0b + seg * A.qstrideb; kbase = A.k0b + seg * A.kstrideb; Lq = A.Lqb; Lk = A.Lkb;
    }
    const int qb = bx * 128;
    if (qb >= Lq) return;
    const int head = by, d = A.d, nct = A.p_nct;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, l31 = lane & 31;
    const int wall = __builtin_amdgcn_readfirstlane(t >> 6), w = wall & 3, kg = wall >> 2;      // query wave, key group (0 when KG = 1)
    const unsigned grp_lds = (unsigned)kg * G::NB * G::BUF;          // the group's tile buffers

    // ---- scales: one per query row, one per key group (uniform over the segment's keys)
    const float kb_bound = A.kgb[kbase], vb_bound = A.vgb[kbase];   // (every key row of a group carries the group's bound)
    const int my_q = qb + w * 32 + l31;
    const bool q_in = my_q < Lq;
    const int qrow = qbase + min(my_q, Lq - 1);
    const bool q_valid = q_in && (!A.qmask || A.qmask[qrow]);
    const float sfac = A.scale * 1.4426950408889634f *
                       __uint_as_float((unsigned)(127 - attn_scale_exp(A.qbnd[qrow]) - attn_scale_exp(kb_bound)) << 23);
    const float vinv = __uint_as_float((unsigned)(127 - attn_scale_exp(vb_bound)) << 23);

    // ---- Q fragments: both planes of the KS chunks of this lane's query, straight from the image
    au32x4 qh[KS], ql[KS];
    {
        const int side = qrow >= A.p_split ? 1 : 0, lrow = qrow - (side ? A.p_split : 0), r = lrow & 127, swq = (r >> 2) & 3;
        const char* qp = A.qimg[side] + (((size_t)(lrow >> 7) * nct + KS * head) * 128 + r) * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qh[s] = *reinterpret_cast<const au32x4*>(qp + (size_t)s * 8192 + ((h ^ swq) << 4));
            ql[s] = *reinterpret_cast<const au32x4*>(qp + (size_t)s * 8192 + (((2 + h) ^ swq) << 4));
        }
    }
    // ---- K / V staging: instruction `ins` of a tile: ins < 2 KS -> K, else V; chunk (ins / 2), rows 16 (ins & 1) + lane / 4
    const int kside = kbase >= A.p_split ? 1 : 0, klrow0 = kbase - (kside ? A.p_split : 0);
    const char* const kimg = A.kimg[kside];
    const char* const vimg = A.vimg[kside];
    // A DMA instruction copies 16 key rows x 64 B of one chunk.  When the segment's first key row is a multiple of 16 inside its image (every
    // loop of the library: key groups start at multiples of 32) and the tile is complete, those 16 rows are ONE contiguous KB of the image:
    // the source is a wave-uniform base (SALU arithmetic, SGPR pair) + 16 bytes per lane -- round 5: the per-lane form (row clamp, block index,
    // 64-bit multiply-adds per lane) was ~12 VALU instructions per DMA instruction, 48 per tile and wave: a fifth of the VALU stream of the
    // d = 64 instantiation, which is VALU-bound (profiles/r05_attention_planes_d64_pmc.json).  Incomplete tiles and odd segments keep it.
    const bool seg_aligned = (klrow0 & 15) == 0;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto stage = [&](int kt, int b) __attribute__((always_inline)) {
        const bool fast = seg_aligned && kt * 32 + 32 <= Lk;
#pragma unroll
        for (int i = 0; i < (4 * KS + 3) / 4; ++i) {
            const int ins = w + 4 * i;
            if (ins < 4 * KS) {
                const bool isv = ins >= 2 * KS;
                const int j = ins - (isv ? 2 * KS : 0), chunk = j >> 1, half = j & 1;
                const unsigned dst = lds_base + grp_lds + b * G::BUF + (isv ? G::KIMG + chunk * G::VCH : chunk * G::KCH) + half * 1024;
                if (fast) {
                    const int lrow0 = klrow0 + kt * 32 + half * 16;
                    const char* sbase = (isv ? vimg : kimg) + (((size_t)(lrow0 >> 7) * nct + KS * head + chunk) * 128 + (lrow0 & 127)) * 64;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(sbase) : "memory");
                } else {
                    const int key = min(kt * 32 + half * 16 + (lane >> 2), Lk - 1), lrow = klrow0 + key;
                    const char* src = (isv ? vimg : kimg) + (((size_t)(lrow >> 7) * nct + KS * head + chunk) * 128 + (lrow & 127)) * 64 + (lane & 3) * 16;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
                }
            }
        }
    };
    // swizzle of the key rows a lane touches: position of logical unit u of row r is u ^ ((r >> 2) & 3); tiles are 32 rows, so
    // it does not depend on the tile
    const int swk = ((klrow0 + l31) >> 2) & 3;                  // K fragment row l31
    // V fragments: lane g16 = l31 & 15 of a 16-lane group supplies row (g16 >> 2) of a 4-key block, feature quarter (g16 & 3)
    const int g16 = l31 & 15, vq = g16 >> 2, vp = g16 & 3;
    unsigned voff[2][2][2];                                      // [k-step][block a / b][hi / lo]: byte offset inside a chunk
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int krow = 16 * s + 8 * blk + 4 * h + vq, sw = ((klrow0 + krow) >> 2) & 3;
            voff[s][blk][0] = krow * 64 + (((vp >> 1) ^ sw) << 4) + (vp & 1) * 8;
            voff[s][blk][1] = krow * 64 + (((2 + (vp >> 1)) ^ sw) << 4) + (vp & 1) * 8;
        }
    int vchunk[NDT];                                             // chunk of feature tile i for this lane (features beyond p_dp: the last chunk again, results unused)
#pragma unroll
    for (int i = 0; i < NDT; ++i) vchunk[i] = min(2 * i + (l31 >> 4), KS - 1) * G::VCH;

    f32x16 acc[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int nkt_all = (Lk + 31) / 32;
    // this group's tiles: absolute tile kg + KG i, i < nkt; the loop runs nkt_max rounds in every group (the barriers are the workgroup's)
    const int nkt = (nkt_all - kg + KG - 1) / KG, nkt_max = (nkt_all + KG - 1) / KG;
    auto tile_of = [&](int i) { return kg + KG * i; };

    // The Q fragments must have ARRIVED before the loop, and the compiler must know it.  Round 5 (ISA of every instantiation): their 2 KS global
    // loads were still "pending" in the compiler's vmcnt model when the loop began, so it placed its counted waits at their first uses -- inside
    // the loop: s_waitcnt vmcnt(2 KS - 1) .. vmcnt(0) down the Q K^T chain of EVERY tile.  vmcnt retires in order and the LDS-DMA of the next tile
    // (inline asm, invisible to that model) is issued just in front of the chain: each tile's chain waited for the whole prefetch it had just
    // issued -- the copy of tile kt + 1 never overlapped the arithmetic of tile kt.  Consuming the registers here puts the one wait here.
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" ::"v"(qh[s]), "v"(ql[s]));

    // (the key mask of a tile is fetched one tile AHEAD too, in front of that tile's DMA, and consumed behind the loop's own wait: loaded inside
    //  the tile, the compiler's wait for it -- vmcnt(0), in order -- again covered the prefetch just issued: cfg3's and every ragged batch's case)
    constexpr int NB = G::NB;
    static_assert((4 * KS) % 4 == 0, "every wave issues KS DMA instructions per tile");
    const bool counted = A.kmask == nullptr;                     // (a mask byte load per tile is a compiler-placed VMEM operation between the DMAs: those
                                                                 //  calls -- 4DMatch, ragged batches -- keep the drain-all wait and a prefetch distance of one)
    unsigned mnext = attn_mask_byte(A, kbase, tile_of(0) * 32, Lk, l31);
#pragma unroll
    for (int j = 0; j < NB - 1; ++j)
        if (j < nkt && (counted || j == 0)) stage(tile_of(j), j);
    for (int kti = 0; kti < nkt_max; ++kti) {
        const int b = kti % NB, kt = tile_of(kti);
        if (KG > 1 && kti >= nkt) {                              // (the other group has one tile more: keep its barrier company)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            continue;
        }
        // own DMA instructions of tile kt have landed; those of the tiles behind it (KS per tile and wave, issued in order) may still fly
        if (counted) {
            const int ahead = min(NB - 2, nkt - 1 - kti);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * KS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(KS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" : "+v"(mnext));                          // (the compiler's wait for this tile's mask byte goes here, where it is free)
        __syncthreads();                                         // ... and everybody's; everybody is done with tile kt - 1's buffer
        const unsigned char mbyte = (unsigned char)mnext;
        if (counted) {
            if (kti + NB - 1 < nkt) stage(tile_of(kti + NB - 1), (kti + NB - 1) % NB);
        } else if (kti + 1 < nkt) {
            mnext = attn_mask_byte(A, kbase, tile_of(kti + 1) * 32, Lk, l31);
            stage(tile_of(kti + 1), (kti + 1) % NB);
        }
        const char* kb = lds + grp_lds + b * G::BUF + l31 * 64;
        const char* vb = lds + grp_lds + b * G::BUF + G::KIMG;
        // ---- S^T = K Q^T
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const ah16x8 kh = *reinterpret_cast<const ah16x8*>(kb + s * G::KCH + ((h ^ swk) << 4));
            const ah16x8 kl = *reinterpret_cast<const ah16x8*>(kb + s * G::KCH + (((2 + h) ^ swk) << 4));
            const ah16x8 q_h = __builtin_bit_cast(ah16x8, qh[s]), q_l = __builtin_bit_cast(ah16x8, ql[s]);
            if (!F16) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, q_h, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, q_l, sc, 0, 0, 0);
            }
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, q_h, sc, 0, 0, 0);
        }
        // ---- mask, scale, running softmax (register r holds key (r&3) + 8(r>>2) + 4h of the tile)
        // The softmax is a serial VALU stretch between the two MFMA blocks of a tile and it does not shrink with the head dim: at d = 64
        // it outweighs the MFMAs.  Two trims: a tile with all 32 keys present and unmasked (wave-uniform test) skips the per-register drop
        // logic and folds the scale into the exponent's argument, p = exp2(fma(s, sfac, -m)) (one rounding instead of two: the last bit
        // of p may differ from the masked path's); and the accumulators are rescaled only when some query's running maximum moved in
        // this tile (alpha = 1 exactly for every lane otherwise).
        const unsigned kbits_all = attn_mask_bits(mbyte);
        const bool full_tile = kt * 32 + 32 <= Lk && kbits_all == 0xFFFFFFFFu;       // (the ballot covers both lane halves: bits 0..31 twice)
        float mx = -INFINITY;
        float alpha = 1.f, psum = 0.f;
        float m_new;
        if (full_tile) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
            mx = attn_xhalf(mx, [](float x, float y) { return fmaxf(x, y); }) * sfac;   // sfac > 0: the maximum commutes with the scale
            m_new = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], sfac, -m_new));
                sc[r] = p;
                psum += p;
            }
        } else {
            const unsigned kbits = kbits_all >> (4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float s = sc[r];
                const bool drop = kk >= Lk || (q_valid && !((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u));     // transformero.py:82
                s = drop ? -INFINITY : s * sfac;
                sc[r] = s;
                mx = fmaxf(mx, s);
            }
            mx = attn_xhalf(mx, [](float x, float y) { return fmaxf(x, y); });
            m_new = fmaxf(m_run, mx);
            if (m_new == -INFINITY) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
            } else {
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sc[r] - m_new);
                    sc[r] = p;
                    psum += p;
                }
            }
        }
        psum = attn_xhalf(psum, [](float x, float y) { return x + y; });
        l_run = l_run * alpha + psum;
        if (__any(m_new != m_run)) {                             // (alpha == 1 exactly in every lane otherwise: exp2(0))
#pragma unroll
            for (int i = 0; i < NDT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        }
        m_run = m_new;
        // ---- O^T += V^T P^T : k-step s contracts the keys of score registers 8 s .. 8 s + 7
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 ph, pl;
            attn_split2(sc[8 * s + 0], sc[8 * s + 1], ph.x, pl.x);
            attn_split2(sc[8 * s + 2], sc[8 * s + 3], ph.y, pl.y);
            attn_split2(sc[8 * s + 4], sc[8 * s + 5], ph.z, pl.z);
            attn_split2(sc[8 * s + 6], sc[8 * s + 7], ph.w, pl.w);
            const au32x4 phv = {ph.x, ph.y, ph.z, ph.w}, plv = {pl.x, pl.y, pl.z, pl.w};
            const ah16x8 p_h = __builtin_bit_cast(ah16x8, phv), p_l = __builtin_bit_cast(ah16x8, plv);
#pragma unroll
            for (int i = 0; i < NDT; ++i) {
                const unsigned base = lds_base + grp_lds + b * G::BUF + G::KIMG + vchunk[i];
                auto tr = [&](unsigned off) {
                    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_lds_s4*)(size_t)(base + off));
                };
                const as16x4 ha = tr(voff[s][0][0]), hb = tr(voff[s][1][0]), la = tr(voff[s][0][1]), lb = tr(voff[s][1][1]);
                typedef short as16x8 __attribute__((ext_vector_type(8)));
                const as16x8 vh8 = {ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
                const as16x8 vl8 = {la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
                const ah16x8 v_h = __builtin_bit_cast(ah16x8, vh8), v_l = __builtin_bit_cast(ah16x8, vl8);
                if (!F16) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_l, p_h, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_h, p_l, acc[i], 0, 0, 0);
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_h, p_h, acc[i], 0, 0, 0);
            }
        }
        (void)vb;
    }
    __syncthreads();

    if constexpr (KG > 1) {
        // ---- merge of the two key groups: group 1 parks (m, l, acc) of its queries, group 0 folds them into its own (lane-wise: both groups
        // hold the same queries in the same lanes)
        float* const mr = smem + (size_t)w * (NDT * 16 + 2) * 64 + lane;
        if (kg == 1) {
            mr[0] = m_run; mr[64] = l_run;
#pragma unroll
            for (int i = 0; i < NDT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mr[(2 + 16 * i + r) * 64] = acc[i][r];
        }
        __syncthreads();
        if (kg == 0) {
            const float m1 = mr[0], l1 = mr[64];
            const float m = fmaxf(m_run, m1);
            const float a0 = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m), a1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - m);
            l_run = l_run * a0 + l1 * a1;
#pragma unroll
            for (int i = 0; i < NDT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = acc[i][r] * a0 + mr[(2 + 16 * i + r) * 64] * a1;
        }
        __syncthreads();                                         // every wave of group 0 has read: the region is free for the output staging
        if (kg == 1) return;                                     // (no workgroup barrier behind this point)
    }

    // ---- out[q][f] = O^T[f][q] 2^-sv / l, through LDS, then the plane image of the merge projection's operand
    float* ob = smem + w * 32 * G::OQS;
    const float inv = vinv / l_run;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[l31 * G::OQS + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[i][r] * inv;
    wave_lds_fence();
    attn_store_planes(A, ob, G::OQS, qbase + qb + w * 32, Lq - (qb + w * 32), head, d, vb_bound, lane);
}


template <int DG, int NDT>
static int configure_attn() {
    using G = AttnGeom<DG, NDT>;
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_kernel<DG, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(G::SMEM_FLOATS * sizeof(float))));
    if constexpr (AttnGeom<DG, NDT, 8>::SMEM_FLOATS * sizeof(float) <= 160 * 1024)
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_kernel<DG, NDT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(AttnGeom<DG, NDT, 8>::SMEM_FLOATS * sizeof(float))));
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_flash_kernel<DG, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(FlashGeom<DG, NDT>::SMEM_FLOATS * sizeof(float))));
    constexpr int KS = (DG * 8 + 15) / 16;
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_flash_split_kernel<KS, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)FlashSplitGeom<KS, NDT>::SMEM));
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_planes_kernel<KS, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)AttnPlGeom<KS, NDT>::SMEM));
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_planes_kernel<KS, NDT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)AttnPlGeom<KS, NDT>::SMEM));
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_planes_kernel<KS, NDT, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)AttnPlGeom<KS, NDT>::SMEM2));
    DR_HIP_CHECK(hipFuncSetAttribute((const void*)attention_planes_kernel<KS, NDT, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)AttnPlGeom<KS, NDT>::SMEM2));

    return DR_OK;
}

// > 64 KiB of dynamic LDS needs a function attribute; set it eagerly (never inside a stream capture)
int attention_configure() {
    int rc = configure_attn<8, 2>();
    if (rc == DR_OK) rc = configure_attn<14, 4>();
    if (rc == DR_OK) rc = configure_attn<17, 5>();
    return rc;
}

static int g_flash_min = -1;    // tests / tools: force the flash form from this many workgroups (-1 = default rule)
static int g_attn_split = -1;   // tests / tools: 0 = f32-input MFMA flash kernel, 1 = split-operand one (-1 = default)
void attention_force_flash_min(int n) { g_flash_min = n; }
void attention_force_split(int on) { g_attn_split = on; }

template <int DG, int NDT>
static int launch_attn(const AttnArgs& a, hipStream_t st) {
    using G = AttnGeom<DG, NDT>;
    const size_t lds = (size_t)G::SMEM_FLOATS * sizeof(float);
    const int maxLq = a.nseg2 > 0 && a.Lqb > a.Lq ? a.Lqb : a.Lq;
    dim3 grid((maxLq + 31) / 32, a.H, a.nseg + a.nseg2);
    // 4 L S C per (segment): QK^T and PV
    const double flops = 4.0 * a.H * a.d * ((double)a.nseg * a.Lq * a.Lk + (double)a.nseg2 * a.Lqb * a.Lkb);
    ProfScope ps(PK_ATTN, flops, st);
    // the flash form has a quarter of the workgroups: below ~one per CU the 32-query kernel fills the chip better
    const int flash_env = env_knob("DR_ATTN_FLASH_MIN", 256);
    const int flash_min = g_flash_min >= 0 ? g_flash_min : flash_env;
    dim3 fgrid((maxLq + 127) / 128, a.H, a.nseg + a.nseg2);
    if (a.qimg[0]) {
        // plane-image operands (the plane path of the loop): fp16 hi / lo products, DMA-fed tiles
        constexpr int KS = (DG * 8 + 15) / 16;
        if (a.p_dp != 16 * KS || !a.pimg[0]) return DR_EINVAL;
        const size_t plds = AttnPlGeom<KS, NDT>::SMEM;
        AttnArgs ax = a;
        // (the dealing is a bijection only when the (head, segment) count is a multiple of 8; one query block per group has nothing to share)
        ax.xcd_groups = ((fgrid.y * fgrid.z) % 8 == 0 && fgrid.x > 1 && env_knob("DR_ATTN_XCD", 1)) ? 1 : 0;
        // at most one workgroup per CU and at least four key tiles: two key groups per workgroup (eight waves, two per SIMD; the groups' partial
        // softmax states merge through LDS).  cfg3's 8-pair call -2.9 % (154.7 -> 159.2 pairs/s), cfg5's -2 %.  Another summation order of the
        // same float32 mathematics: every soft-family bound stays a plain 1e-4; on the stress family it moves WHICH sharp entries sit near their
        // 2 x rule (an earlier tree had one of test_cfg3_4dmatch_512_batch8_20_steps cross it; on this tree the whole suite is green with it on,
        // profiles/r06_cfg3_experiments.json).  DR_ATTN_KG2=0 (diagnostics) switches it off.
        const int minLk = a.nseg2 > 0 && a.Lkb < a.Lk ? a.Lkb : a.Lk;
        const bool kg2 = (long)fgrid.x * fgrid.y * fgrid.z <= (long)device_cu_count() && minLk >= 128 && env_knob("DR_ATTN_KG2", 1) != 0;
        if (kg2) {
            const size_t plds2 = AttnPlGeom<KS, NDT>::SMEM2;
            if (a.f16_single) hipLaunchKernelGGL((attention_planes_kernel<KS, NDT, true, 2>), fgrid, dim3(512), plds2, st, ax);
            else hipLaunchKernelGGL((attention_planes_kernel<KS, NDT, false, 2>), fgrid, dim3(512), plds2, st, ax);
        } else if (a.f16_single) hipLaunchKernelGGL((attention_planes_kernel<KS, NDT, true>), fgrid, dim3(256), plds, st, ax);
        else hipLaunchKernelGGL((attention_planes_kernel<KS, NDT>), fgrid, dim3(256), plds, st, ax);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    if (a.pimg[0] || ((int)(fgrid.x * fgrid.y * fgrid.z) >= flash_min && (a.ldo % 4) == 0 && (((uintptr_t)a.out) & 15) == 0)) {
        const int split_env = env_knob("DR_ATTN_SPLIT", 1);
        const int mode = g_attn_split >= 0 ? g_attn_split : split_env;     // 0 f32-input MFMA, 1 split operands
        // d = 132 (4DMatch) needs 288 V staging blocks and 9 k-steps of Q in registers: more than two waves per SIMD
        // allow -> f32 kernel.  (A hybrid with only PV split measured slower than both: 93 vs 84 / 108 us.)
        if (DG <= 14 && mode == 1) {
            constexpr int KS = (DG * 8 + 15) / 16;
            using SG = FlashSplitGeom<KS, NDT>;
            const size_t slds = (size_t)SG::SMEM;
            hipLaunchKernelGGL((attention_flash_split_kernel<KS, NDT>), fgrid, dim3(256), slds, st, a);
            DR_LAUNCH_CHECK();
            return DR_OK;
        }
        using FG = FlashGeom<DG, NDT>;
        const size_t flds = (size_t)FG::SMEM_FLOATS * sizeof(float);
        hipLaunchKernelGGL((attention_flash_kernel<DG, NDT>), fgrid, dim3(256), flds, st, a);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    // few workgroups (a single pair: 8 query tiles x heads x sides) and a long key range: 8 waves deal the key tiles, a wave's chain
    // of dependent MFMAs (120 of 64 cycles per key tile at d = 108) is then half as long
    constexpr size_t lds8 = (size_t)AttnGeom<DG, NDT, 8>::SMEM_FLOATS * sizeof(float);
    if constexpr (lds8 <= 160 * 1024) {
        const int w8_env = env_knob("DR_ATTN_W8_MAX", 256);
        const int maxLk = a.nseg2 > 0 && a.Lkb > a.Lk ? a.Lkb : a.Lk;
        if ((int)(grid.x * grid.y * grid.z) <= w8_env && maxLk > 4 * 32) {
            hipLaunchKernelGGL((attention_kernel<DG, NDT, 8>), grid, dim3(512), lds8, st, a);
            DR_LAUNCH_CHECK();
            return DR_OK;
        }
    }
    hipLaunchKernelGGL((attention_kernel<DG, NDT>), grid, dim3(256), lds, st, a);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_attention(const AttnArgs& a, hipStream_t st) {
    if (a.d % 4 || (!a.qimg[0] && (a.ldq % 4 || a.ldk % 4 || a.ldv % 4))) return DR_ENOSUP;
    if (a.nseg + a.nseg2 <= 0) return DR_OK;
    if (a.d <= 64) return launch_attn<8, 2>(a, st);       // 2D-3D: d = 64
    if (a.d <= 112) return launch_attn<14, 4>(a, st);     // 3DMatch: d = 108
    if (a.d <= 136) return launch_attn<17, 5>(a, st);     // 4DMatch: d = 132
    return DR_ENOSUP;
}

}  // namespace dr
