// sinkhorn.hip -- batched Sinkhorn normalisation with dustbins for gfx950.
//
// Replaces log_optimal_transport + exp + slice of the reference (3D/models/matching.py:61-93 and
// its three call sites, see include/diffreg_hip.h).  Two kernels:
//
//  sk_reg_kernel     one workgroup per tile, the whole tile (<= 256 x 256) lives in VGPRs as
//                    row-max-shifted exponentials E_ij = exp(Z_ij - rho_i) in [0,1].  The log-domain
//                    iteration  u_i = log mu_i - LSE_j(Z_ij + v_j),  v_j = log nu_j - LSE_i(Z_ij + u_i)
//                    is run in its algebraically identical scaling form
//                        a_i = mu_i / sum_j E_ij b_j ,   b_j = nu_j / sum_i E_ij a_i
//                    (a_i = exp(u_i + rho_i), b_j = exp(v_j)), so each pass is one FMA per element
//                    instead of one exp.  The dustbin row/column (score alpha everywhere) keep every
//                    sum strictly positive, so the form is overflow/underflow safe for any finite
//                    or -inf scores.  HBM traffic = read the tile once, write it once.
//                    Row sums: 16 rows per wave, 4 columns per lane -> one butterfly reduce-scatter
//                    (17 shuffles for 16 rows).  Column sums: per-wave partials through LDS.
//
//  sk_stream_kernel  any size / strict input dtype / log output: E is kept in a global workspace
//                    (L2/MALL resident), one workgroup per tile, same scaling iteration.
#include "kernels.h"
#include <type_traits>
#include <stdlib.h>

namespace dr {

struct SkArgs {
    const void* scores;
    const uint8_t* src_mask;
    const uint8_t* tgt_mask;
    const float* bin_score;
    void* out;
    void* ws;
    const double* shift;   // per-tile value subtracted from the scores on load (nullable)
    int B, N, M, iters, flags, vec_in, vec_out;
    unsigned spin_limit;   // co-resident form: polls a workgroup makes before it gives up (dr_device_status)
    unsigned* call_status; // nullable: the CALLER's own sticky word (a loop call's workspace): bit 0 is set beside the process-wide flag
};

// ---------------------------------------------------------------------------------------------
// reduce-scatter of 16 per-lane values over the 64 lanes of a wave: on return lane l holds the
// reduction of p[(l >> 2) & 15] over all lanes.
// ---------------------------------------------------------------------------------------------
template <typename Op>
__device__ __forceinline__ float reduce16(const float (&p)[16], Op op) {
    const int l = lane_id();
    const bool b5 = l & 32, b4 = l & 16, b3 = l & 8, b2 = l & 4;
    float q8[8], q4[4], q2[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float send = b5 ? p[k] : p[k + 8];
        float keep = b5 ? p[k + 8] : p[k];
        q8[k] = op(keep, __shfl_xor(send, 32));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float send = b4 ? q8[k] : q8[k + 4];
        float keep = b4 ? q8[k + 4] : q8[k];
        q4[k] = op(keep, __shfl_xor(send, 16));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float send = b3 ? q4[k] : q4[k + 2];
        float keep = b3 ? q4[k + 2] : q4[k];
        q2[k] = op(keep, __shfl_xor(send, 8));
    }
    float send = b2 ? q2[0] : q2[1];
    float keep = b2 ? q2[1] : q2[0];
    float r = op(keep, __shfl_xor(send, 4));
    r = op(r, __shfl_xor(r, 2));
    r = op(r, __shfl_xor(r, 1));
    return r;
}

struct OpAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };

__device__ __forceinline__ float bcast_lane(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// streaming (non-temporal) global access of a lane's CPL values: a tile is read once and written once, so the big batched
// launches mark both as streaming -- on MI355X a 1024-thread tile copy runs at 6.15 TB/s with `nt` on loads and stores
// against 5.67 TB/s without (tools/_build/skx.hip).  Small launches keep the default policy: their consumer is the next
// kernel of the loop and the tile is still in L2.
typedef float sk_v4f __attribute__((ext_vector_type(4)));
typedef float sk_v2f __attribute__((ext_vector_type(2)));
typedef double sk_v2d __attribute__((ext_vector_type(2)));
template <typename T, int CPL>
struct NtIO;
template <>
struct NtIO<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4], double sh) {
        const sk_v4f t = __builtin_nontemporal_load(reinterpret_cast<const sk_v4f*>(p));
        const float s = (float)sh;
        v[0] = t.x - s; v[1] = t.y - s; v[2] = t.z - s; v[3] = t.w - s;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        sk_v4f t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        __builtin_nontemporal_store(t, reinterpret_cast<sk_v4f*>(p));
    }
};
template <>
struct NtIO<double, 4> {
    static __device__ __forceinline__ void load(const double* p, float (&v)[4], double sh) {
        const sk_v2d t0 = __builtin_nontemporal_load(reinterpret_cast<const sk_v2d*>(p));
        const sk_v2d t1 = __builtin_nontemporal_load(reinterpret_cast<const sk_v2d*>(p + 2));
        v[0] = (float)(t0.x - sh); v[1] = (float)(t0.y - sh); v[2] = (float)(t1.x - sh); v[3] = (float)(t1.y - sh);
    }
    static __device__ __forceinline__ void store(double* p, const float (&v)[4]) {
        sk_v2d t0, t1; t0.x = (double)v[0]; t0.y = (double)v[1]; t1.x = (double)v[2]; t1.y = (double)v[3];
        __builtin_nontemporal_store(t0, reinterpret_cast<sk_v2d*>(p));
        __builtin_nontemporal_store(t1, reinterpret_cast<sk_v2d*>(p + 2));
    }
};
template <>
struct NtIO<float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[2], double sh) {
        const sk_v2f t = __builtin_nontemporal_load(reinterpret_cast<const sk_v2f*>(p));
        const float s = (float)sh;
        v[0] = t.x - s; v[1] = t.y - s;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) {
        sk_v2f t; t.x = v[0]; t.y = v[1];
        __builtin_nontemporal_store(t, reinterpret_cast<sk_v2f*>(p));
    }
};
template <>
struct NtIO<double, 2> {
    static __device__ __forceinline__ void load(const double* p, float (&v)[2], double sh) {
        const sk_v2d t = __builtin_nontemporal_load(reinterpret_cast<const sk_v2d*>(p));
        v[0] = (float)(t.x - sh); v[1] = (float)(t.y - sh);
    }
    static __device__ __forceinline__ void store(double* p, const float (&v)[2]) {
        sk_v2d t; t.x = (double)v[0]; t.y = (double)v[1];
        __builtin_nontemporal_store(t, reinterpret_cast<sk_v2d*>(p));
    }
};

template <typename T, int CPL>
struct VecIO;
template <>
struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4], double sh) {
        float4 t = *reinterpret_cast<const float4*>(p);
        const float s = (float)sh;
        v[0] = t.x - s; v[1] = t.y - s; v[2] = t.z - s; v[3] = t.w - s;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct VecIO<double, 4> {
    static __device__ __forceinline__ void load(const double* p, float (&v)[4], double sh) {
        double2 t0 = *reinterpret_cast<const double2*>(p);
        double2 t1 = *reinterpret_cast<const double2*>(p + 2);
        v[0] = (float)(t0.x - sh); v[1] = (float)(t0.y - sh); v[2] = (float)(t1.x - sh); v[3] = (float)(t1.y - sh);
    }
    static __device__ __forceinline__ void store(double* p, const float (&v)[4]) {
        *reinterpret_cast<double2*>(p) = make_double2((double)v[0], (double)v[1]);
        *reinterpret_cast<double2*>(p + 2) = make_double2((double)v[2], (double)v[3]);
    }
};
// fp16 storage (dr_sinkhorn_f16: opt-in, half the HBM bytes of the fp32 tile; the arithmetic stays fp32)
typedef _Float16 sk_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 sk_h2 __attribute__((ext_vector_type(2)));
template <>
struct VecIO<_Float16, 4> {
    static __device__ __forceinline__ void load(const _Float16* p, float (&v)[4], double sh) {
        const sk_h4 t = *reinterpret_cast<const sk_h4*>(p);
        const float s = (float)sh;
        v[0] = (float)t.x - s; v[1] = (float)t.y - s; v[2] = (float)t.z - s; v[3] = (float)t.w - s;
    }
    static __device__ __forceinline__ void store(_Float16* p, const float (&v)[4]) {
        sk_h4 t; t.x = (_Float16)v[0]; t.y = (_Float16)v[1]; t.z = (_Float16)v[2]; t.w = (_Float16)v[3];
        *reinterpret_cast<sk_h4*>(p) = t;
    }
};
template <>
struct VecIO<_Float16, 2> {
    static __device__ __forceinline__ void load(const _Float16* p, float (&v)[2], double sh) {
        const sk_h2 t = *reinterpret_cast<const sk_h2*>(p);
        const float s = (float)sh;
        v[0] = (float)t.x - s; v[1] = (float)t.y - s;
    }
    static __device__ __forceinline__ void store(_Float16* p, const float (&v)[2]) {
        sk_h2 t; t.x = (_Float16)v[0]; t.y = (_Float16)v[1];
        *reinterpret_cast<sk_h2*>(p) = t;
    }
};
template <>
struct VecIO<float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[2], double sh) {
        float2 t = *reinterpret_cast<const float2*>(p);
        const float s = (float)sh;
        v[0] = t.x - s; v[1] = t.y - s;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    }
};
template <>
struct VecIO<double, 2> {
    static __device__ __forceinline__ void load(const double* p, float (&v)[2], double sh) {
        double2 t = *reinterpret_cast<const double2*>(p);
        v[0] = (float)(t.x - sh); v[1] = (float)(t.y - sh);
    }
    static __device__ __forceinline__ void store(double* p, const float (&v)[2]) {
        *reinterpret_cast<double2*>(p) = make_double2((double)v[0], (double)v[1]);
    }
};

// ---------------------------------------------------------------------------------------------
// register-resident kernel: NW waves x 16 rows, 64 lanes x CPL columns
// ---------------------------------------------------------------------------------------------
template <typename TIn, typename TOut, int NW, int CPL, bool VEC>
__global__ __launch_bounds__(NW * 64) void sk_reg_kernel(SkArgs A) {
    constexpr int RPW = 16;
    constexpr int MAXC = 64 * CPL;
    __shared__ __attribute__((aligned(16))) float s_colpart[NW][MAXC];
    __shared__ __attribute__((aligned(16))) float s_b[MAXC + 4];
    __shared__ float s_dust[NW];
    __shared__ float s_red[NW];

    const int N = A.N, M = A.M;
    const int tile = blockIdx.x;
    const int lane = lane_id(), w = wave_id(), t = threadIdx.x;
    const TIn* src = reinterpret_cast<const TIn*>(A.scores) + (size_t)tile * N * M;
    TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * N * M;
    const float alpha = *A.bin_score;
    const int row0 = w * RPW, col0 = lane * CPL;
    const double sh = A.shift ? A.shift[tile] : 0.0;

    // ---- issue all tile loads first ----------------------------------------------------------
    float E[RPW][CPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
#pragma unroll
        for (int c = 0; c < CPL; ++c) E[r][c] = -INFINITY;
        if (row < N) {
            if (VEC) {
                if (col0 < M) VecIO<TIn, CPL>::load(src + (unsigned)(row * M + col0), E[r], sh);
            } else {
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    if (col0 + c < M) E[r][c] = (float)(src[(unsigned)(row * M + col0 + c)] - (TIn)sh);
            }
        }
    }

    // ---- mask sums (ms, ns) and per-thread validity -------------------------------------------
    int ms = N, ns = M;
    if (A.src_mask) ms = __syncthreads_count(t < N && A.src_mask[(size_t)tile * N + t]);
    if (A.tgt_mask) ns = __syncthreads_count(t < M && A.tgt_mask[(size_t)tile * M + t]);
    bool cvalid[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int col = col0 + c;
        cvalid[c] = col < M && (!(A.flags & DR_SK_APPLY_MASK) || !A.tgt_mask || A.tgt_mask[(size_t)tile * M + col]);
    }
    unsigned rvalid = 0;   // bit r: row in range and not masked
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        const bool ok = row < N && (!(A.flags & DR_SK_APPLY_MASK) || !A.src_mask || A.src_mask[(size_t)tile * N + row]);
        rvalid |= (ok ? 1u : 0u) << r;
    }

    // ---- x - min(x) over the raw tile (3D/models/pipeline.py:239,264) --------------------------
    if (A.flags & DR_SK_MINSHIFT) {
        float mn = INFINITY;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (row0 + r < N && col0 + c < M) mn = fminf(mn, E[r][c]);
        mn = wave_min(mn);
        if (lane == 0) s_red[w] = mn;
        __syncthreads();
        mn = s_red[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) mn = fminf(mn, s_red[k]);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int c = 0; c < CPL; ++c) E[r][c] -= mn;
    }

    // ---- mask, row maxima, exponentials --------------------------------------------------------
    float pm[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (!(((rvalid >> r) & 1u) && cvalid[c])) E[r][c] = -INFINITY;
            m = fmaxf(m, E[r][c]);
        }
        pm[r] = m;
    }
    // lane-distributed per-row scalars: lane l holds the value of row row0 + ((l >> 2) & 15);
    // readlane(x, 4r) broadcasts row r into an SGPR when a whole-wave multiplier is needed.
    const bool ragged = (A.flags & DR_SK_RAGGED) != 0;      // padded rows / columns do not exist (no mass, no dustbin share)
    const bool my_row_in = row0 + ((lane >> 2) & 15) < N && (!ragged || ((rvalid >> ((lane >> 2) & 15)) & 1u));
    const float rho_l = fmaxf(alpha, reduce16(pm, OpMax()));
    constexpr float LOG2E = 1.4426950408889634f;
    // exp(alpha - rho_r): the dustbin-column entry of each row
    const float ed_l = my_row_in ? exp2f((alpha - rho_l) * LOG2E) : 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float nrho = -bcast_lane(rho_l, 4 * r) * LOG2E;
#pragma unroll
        for (int c = 0; c < CPL; ++c) E[r][c] = exp2f(fmaf(E[r][c], LOG2E, nrho));   // exp(-inf) = 0
    }

    // ---- marginals (3D/models/matching.py:79-82; padded rows/cols keep mass, quirk Q19) ---------
    const float tot = (float)(ms + ns);
    const float mu = 1.f / tot, muN = (float)ns / tot, nu = mu, nuM = (float)ms / tot;

    float bj[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) bj[c] = (col0 + c < M && (!ragged || cvalid[c])) ? 1.f : 0.f;
    const bool tcol_in = t < M && (!ragged || !A.tgt_mask || A.tgt_mask[(size_t)tile * M + (t < M ? t : 0)]);
    float bM = 1.f;
    float a_l = 0.f;
    float aN = 0.f;

    for (int it = 0; it < A.iters; ++it) {
        // row pass: a_i = mu / (sum_j E_ij b_j + ed_i b_M)
        float p[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) s = fmaf(E[r][c], bj[c], s);
            p[r] = s;
        }
        const float rs_l = reduce16(p, OpAdd());
        a_l = my_row_in ? mu / fmaf(ed_l, bM, rs_l) : 0.f;
        float bs = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) bs += bj[c];
        bs = wave_sum(bs);
        aN = muN / (bs + bM);
        // column pass: b_j = nu / (sum_i E_ij a_i + a_N)
        float cp[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) cp[c] = 0.f;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float ar = bcast_lane(a_l, 4 * r);
#pragma unroll
            for (int c = 0; c < CPL; ++c) cp[c] = fmaf(E[r][c], ar, cp[c]);
        }
        const float dp = wave_sum((lane & 3) == 0 ? ed_l * a_l : 0.f);
        VecIO<float, CPL>::store(&s_colpart[w][col0], cp);
        if (lane == 0) s_dust[w] = dp;
        __syncthreads();
        if (t < MAXC) {
            float c = aN;
#pragma unroll
            for (int k = 0; k < NW; ++k) c += s_colpart[k][t];
            s_b[t] = tcol_in ? nu / c : 0.f;
        } else if (t == MAXC) {
            float c = aN;
#pragma unroll
            for (int k = 0; k < NW; ++k) c += s_dust[k];
            s_b[MAXC] = nuM / c;
        }
        __syncthreads();
        VecIO<float, CPL>::load(&s_b[col0], bj, 0.0);
        bM = s_b[MAXC];
    }

    // ---- conf_ij = E_ij a_i b_j exp(-norm)  (exp + [:-1,:-1] slice of matching.py:215-216) -----
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= N) continue;
        float o[CPL];
        const float ar = bcast_lane(a_l, 4 * r) * tot;
#pragma unroll
        for (int c = 0; c < CPL; ++c) o[c] = E[r][c] * ar * bj[c];
        if (VEC) {
            if (col0 < M) VecIO<TOut, CPL>::store(dst + (unsigned)(row * M + col0), o);
        } else {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (col0 + c < M) dst[(unsigned)(row * M + col0 + c)] = (TOut)o[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fast register-resident kernel: FULL tiles (N = 16 NW, M = 64 CPL), no masks, shift by pointer --
// the shape of every Sinkhorn call of the 3D loop at N = M = 256 / 128.  Same mathematics as
// sk_reg_kernel with every predicate removed, raw v_exp_f32, and cross-lane reductions on
// v_permlane32_swap / v_permlane16_swap / DPP row rotations instead of ds_bpermute.
// Per-row scalars are held 4 per lane: lane l owns rows 4 (l >> 4) + k, k = 0..3 of its wave.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32(float& a, float& b) {   // lanes 32-63 of a <-> lanes 0-31 of b
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {   // odd 16-lane rows of a <-> even rows of b
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int ROR>
__device__ __forceinline__ float row_ror(float v) {            // rotate right inside each 16-lane row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + ROR, 0xF, 0xF, false));
}
template <typename Op>
__device__ __forceinline__ float row_allreduce(float v, Op op) {
    v = op(v, row_ror<8>(v));
    v = op(v, row_ror<4>(v));
    v = op(v, row_ror<2>(v));
    v = op(v, row_ror<1>(v));
    return v;
}
// 16 per-lane values -> q[k] = reduction over the wave of p[4 (lane >> 4) + k]
template <typename Op>
__device__ __forceinline__ void reduce16x4(float (&p)[16], float (&q)[4], Op op) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { swap32(p[k], p[k + 8]); p[k] = op(p[k], p[k + 8]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { swap16(p[k], p[k + 4]); p[k] = op(p[k], p[k + 4]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = row_allreduce(p[k], op);
}
__device__ __forceinline__ float wave_allsum_dpp(float v) {
    v = row_allreduce(v, OpAdd());
    return bcast_lane(v, 0) + bcast_lane(v, 16) + bcast_lane(v, 32) + bcast_lane(v, 48);
}

// launches of at least this many 256 x 256 tiles (128 MB in + 128 MB out: beyond the L2s and half the Infinity Cache)
// stream their tiles with non-temporal loads and stores
constexpr int SK_NT_MIN_TILES = 512;

// The tile lives in registers as float2 pairs of adjacent columns so that every pass is v_pk_fma_f32 / v_pk_mul_f32 (the
// full-rate fp32 form of gfx950: two columns per lane per instruction): with one workgroup per CU nothing overlaps a tile's
// compute phase with memory, so its instruction count is wall time of the batched launch.
typedef float sk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sk_f2 sk_fma2(sk_f2 a, sk_f2 b, sk_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ sk_f2 sk_splat(float v) { sk_f2 r; r.x = v; r.y = v; return r; }

template <typename IO, typename T, int CPL>
__device__ __forceinline__ void sk_load_row(const T* p, sk_f2 (&e)[CPL / 2], double sh) {
    float tmp[CPL];
    IO::load(p, tmp, sh);
#pragma unroll
    for (int k = 0; k < CPL / 2; ++k) { e[k].x = tmp[2 * k]; e[k].y = tmp[2 * k + 1]; }
}

// Everything between "the tile is in registers" and "a_i, b_j are final": row maxima, exponentials, `iters` scaling
// iterations.  `after_max()` runs once the raw scores have been consumed (the persistent kernel issues its prefetch there).
template <int NW, int CPL, typename Hook>
__device__ __forceinline__ void sk_fast_tile(sk_f2 (&E)[16][CPL / 2], const float alpha, const int iters, float (*s_colpart)[64 * CPL],
                                             float* s_b, float* s_dust, float (&a4)[4], sk_f2 (&bj)[CPL / 2], Hook&& after_max) {
    constexpr int RPW = 16, N = NW * RPW, M = 64 * CPL, H = CPL / 2;
    constexpr float LOG2E = 1.4426950408889634f;
    const int lane = lane_id(), w = wave_id(), t = threadIdx.x;
    float p[RPW], rho4[4], ed4[4];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float m = fmaxf(E[r][0].x, E[r][0].y);
#pragma unroll
        for (int k = 1; k < H; ++k) m = fmaxf(m, fmaxf(E[r][k].x, E[r][k].y));
        p[r] = m;
    }
    after_max();
    reduce16x4(p, rho4, OpMax());
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rho4[k] = fmaxf(alpha, rho4[k]);
        ed4[k] = __builtin_amdgcn_exp2f((alpha - rho4[k]) * LOG2E);     // dustbin-column entry of the row
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const sk_f2 nrho = sk_splat(-bcast_lane(rho4[r & 3], 16 * (r >> 2)) * LOG2E);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const sk_f2 z = sk_fma2(E[r][k], sk_splat(LOG2E), nrho);
            E[r][k].x = __builtin_amdgcn_exp2f(z.x);
            E[r][k].y = __builtin_amdgcn_exp2f(z.y);
        }
    }
    const float tot = (float)(N + M);
    const float mu = 1.f / tot, muN = (float)M / tot, nu = mu, nuM = (float)N / tot;
#pragma unroll
    for (int k = 0; k < H; ++k) bj[k] = sk_splat(1.f);
    float bM = 1.f, aN = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            sk_f2 s = E[r][0] * bj[0];
#pragma unroll
            for (int k = 1; k < H; ++k) s = sk_fma2(E[r][k], bj[k], s);
            p[r] = s.x + s.y;
        }
        float rs4[4];
        reduce16x4(p, rs4, OpAdd());
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[k] = mu * __builtin_amdgcn_rcpf(fmaf(ed4[k], bM, rs4[k]));      // v_rcp_f32 (1 ulp) instead of the ~9-instruction IEEE division
        sk_f2 bs2 = bj[0];
#pragma unroll
        for (int k = 1; k < H; ++k) bs2 += bj[k];
        aN = muN * __builtin_amdgcn_rcpf(wave_allsum_dpp(bs2.x + bs2.y) + bM);
        sk_f2 cp[H];
#pragma unroll
        for (int k = 0; k < H; ++k) cp[k] = sk_splat(0.f);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const sk_f2 ar = sk_splat(bcast_lane(a4[r & 3], 16 * (r >> 2)));
#pragma unroll
            for (int k = 0; k < H; ++k) cp[k] = sk_fma2(E[r][k], ar, cp[k]);
        }
        float dpl = ed4[0] * a4[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) dpl = fmaf(ed4[k], a4[k], dpl);
        const float dp = bcast_lane(dpl, 0) + bcast_lane(dpl, 16) + bcast_lane(dpl, 32) + bcast_lane(dpl, 48);
        float cpf[CPL];
#pragma unroll
        for (int k = 0; k < H; ++k) { cpf[2 * k] = cp[k].x; cpf[2 * k + 1] = cp[k].y; }
        VecIO<float, CPL>::store(&s_colpart[w][lane * CPL], cpf);
        if (lane == 0) s_dust[w] = dp;
        __syncthreads();
        if (t < M) {
            float c = aN;
#pragma unroll
            for (int k = 0; k < NW; ++k) c += s_colpart[k][t];
            s_b[t] = nu * __builtin_amdgcn_rcpf(c);
        } else if (t == M) {
            float c = aN;
#pragma unroll
            for (int k = 0; k < NW; ++k) c += s_dust[k];
            s_b[M] = nuM * __builtin_amdgcn_rcpf(c);
        }
        __syncthreads();
        float bf[CPL];
        VecIO<float, CPL>::load(&s_b[lane * CPL], bf, 0.0);
#pragma unroll
        for (int k = 0; k < H; ++k) { bj[k].x = bf[2 * k]; bj[k].y = bf[2 * k + 1]; }
        bM = s_b[M];
    }
}

// out row r = E[r] a_r b (N + M)
template <typename IO, typename T, int CPL>
__device__ __forceinline__ void sk_store_row(T* p, const sk_f2 (&e)[CPL / 2], const float ar_tot, const sk_f2 (&bj)[CPL / 2]) {
    float o[CPL];
#pragma unroll
    for (int k = 0; k < CPL / 2; ++k) {
        const sk_f2 v = e[k] * sk_splat(ar_tot) * bj[k];
        o[2 * k] = v.x; o[2 * k + 1] = v.y;
    }
    IO::store(p, o);
}

template <typename TIn, typename TOut, int NW, int CPL, bool NT>
__global__ __launch_bounds__(NW * 64) void sk_fast_kernel(SkArgs A) {
    using InIO = typename std::conditional<NT, NtIO<TIn, CPL>, VecIO<TIn, CPL>>::type;
    using OutIO = typename std::conditional<NT, NtIO<TOut, CPL>, VecIO<TOut, CPL>>::type;
    constexpr int RPW = 16, N = NW * RPW, M = 64 * CPL;
    __shared__ __attribute__((aligned(16))) float s_colpart[NW][M];
    __shared__ __attribute__((aligned(16))) float s_b[M + 4];
    __shared__ float s_dust[NW];

    const int tile = blockIdx.x, lane = lane_id(), w = wave_id();
    const TIn* src = reinterpret_cast<const TIn*>(A.scores) + (size_t)tile * N * M + (unsigned)(w * RPW * M + lane * CPL);
    TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * N * M + (unsigned)(w * RPW * M + lane * CPL);
    const float alpha = *A.bin_score;
    const double sh = A.shift ? A.shift[tile] : 0.0;

    sk_f2 E[RPW][CPL / 2], bj[CPL / 2];
    float a4[4];
#pragma unroll
    for (int r = 0; r < RPW; ++r) sk_load_row<InIO, TIn, CPL>(src + r * M, E[r], sh);
    sk_fast_tile<NW, CPL>(E, alpha, A.iters, s_colpart, s_b, s_dust, a4, bj, [] {});
    const float tot = (float)(N + M);
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        sk_store_row<OutIO, TOut, CPL>(dst + r * M, E[r], bcast_lane(a4[r & 3], 16 * (r >> 2)) * tot, bj);
}

// ---------------------------------------------------------------------------------------------
// persistent form of the fast kernel for big batches of 256 x 256 float tiles (the roofline micro-benchmark's regime):
// a workgroup holds one tile in 64 VGPRs per thread, so only ONE workgroup fits a CU and a tile's compute phase
// (exponentials + 3 iterations, ~6 us) can overlap memory traffic only if this workgroup itself keeps traffic in flight.
// A workgroup walks over tiles g, g + G, ...; EVERY row of the next tile arrives through a 128 KB LDS image (8 rows per wave)
// by LDS-DMA, in two halves, and every global access of the loop is hand-placed asm with counted waits:
//   compute(tile i)   ... the wave's rows 0..7 of tile i + 1 land in the image (DMA issued once the raw scores are consumed)
//   stores rows 0..7 | vmcnt(8): that DMA has landed | image -> registers of rows 0..7 | DMA rows 8..15 of tile i + 1 |
//   stores rows 8..15 | vmcnt(8): the DMA and the first stores have landed | image -> registers of rows 8..15 | compute(i + 1)
// vmcnt retires in order, so a wave can only learn that a load has landed after everything it issued before it has: the last 8
// stores of a tile are issued BEHIND the last loads and drain under the next tile's compute phase together with its prefetch
// (256 KB of the 512 KB a tile moves), and nothing in the loop is a compiler-placed load (round 2's form re-loaded half the
// tile into registers behind the stores: hipcc answered with vmcnt(0) in front of the compute phase, which also waited for
// the just-issued prefetch: the whole 512 KB was exposed, `profiles/r03_sinkhorn_persist_isa_waits.txt`).
// ---------------------------------------------------------------------------------------------
// workgroups of the persistent form: one per CU; DR_SK_PERSIST_GRID=0 keeps the one-tile-per-workgroup kernel (tools)
static int sk_persist_grid() {
    const int v = env_knob("DR_SK_PERSIST_GRID", 256);
    return v;
}
typedef __attribute__((address_space(3))) void sk_lds_void;
typedef unsigned sk_u4 __attribute__((ext_vector_type(4)));
constexpr int SKP_PRE = 8;                                        // rows per wave in the LDS image
constexpr size_t SKP_LDS = 16 * 256 * 4 + 1040 + 64 + 16 * SKP_PRE * 1024;    // colpart + b + dust + image

// 1 KB per wave-instruction, SGPR base + VGPR byte offset + immediate (added to the global AND to the LDS address M0 + 16 lane)
#define SK_DMA(ldsaddr, voff, gbase, imm)                                                                      \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3 nt" ::"s"(ldsaddr), "v"(voff), "s"(gbase), "i"(imm) : "memory")
// (s_nop 1 behind the store: a VMEM store of more than 64 bits must not be followed at once by a VALU write of its data registers --
//  hipcc's hazard recognizer pads its own stores, it cannot see into an asm block; without it the next row's products, computed into
//  the same registers, raced the store's data read: run-to-run different tiles)
#define SK_STORE(voff, data, gbase, imm) \
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" ::"v"(voff), "v"(data), "s"(gbase), "i"(imm) : "memory")

// the wave's 8 image rows -> registers (one block: the results are valid when it ends)
__device__ __forceinline__ void skp_read_image(unsigned addr, sk_f2 (*E)[2]) {
    sk_u4 r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
        "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(addr)
        : "memory");
    const sk_u4 rr[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        E[r][0].x = __uint_as_float(rr[r].x); E[r][0].y = __uint_as_float(rr[r].y);
        E[r][1].x = __uint_as_float(rr[r].z); E[r][1].y = __uint_as_float(rr[r].w);
    }
}

__global__ __launch_bounds__(1024) void sk_fast_persist_kernel(SkArgs A) {
    constexpr int NW = 16, CPL = 4, RPW = 16, N = 256, M = 256;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*s_colpart)[M] = reinterpret_cast<float (*)[M]>(sk_smem);
    float* s_b = reinterpret_cast<float*>(sk_smem + NW * M * 4);
    float* s_dust = reinterpret_cast<float*>(sk_smem + NW * M * 4 + 1040);
    char* s_pref = sk_smem + NW * M * 4 + 1040 + 64;

    const int lane = lane_id(), w = wave_id();
    const unsigned toff = (unsigned)(w * RPW * M + lane * CPL);
    const float alpha = *A.bin_score;
    const float tot = (float)(N + M);
    // this wave's 8 image rows: wave-uniform LDS base (M0 of the DMA) and the lane's read address
    const unsigned img = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(sk_lds_void*)(s_pref + w * SKP_PRE * 1024));
    const unsigned img_lane = img + (unsigned)lane * 16;
    const unsigned voff = toff * 4;                               // byte offset of the lane's float4 in row 0 of the wave
    const float* scores = reinterpret_cast<const float*>(A.scores);

    sk_f2 E[RPW][CPL / 2], bj[CPL / 2];
    float a4[4];
    int tile = blockIdx.x;
    // rows 8 half .. 8 half + 7 of the tile at `base` -> image rows 0 .. 7
    auto dma_half = [&](const char* base, int half) __attribute__((always_inline)) {
        const unsigned v0 = voff + half * 8192;
        SK_DMA(img, v0, base, 0); SK_DMA(img, v0, base, 1024); SK_DMA(img, v0, base, 2048); SK_DMA(img, v0, base, 3072);
        SK_DMA(img + 4096, v0 + 4096, base, 0); SK_DMA(img + 4096, v0 + 4096, base, 1024);
        SK_DMA(img + 4096, v0 + 4096, base, 2048); SK_DMA(img + 4096, v0 + 4096, base, 3072);
    };
    {   // the first tile takes the same road (no compiler-placed load anywhere in this kernel: hipcc's own vmcnt waits for one
        // would sit in the loop header and wait for the stores and prefetches in flight there)
        const char* base0 = reinterpret_cast<const char*>(scores + (size_t)tile * N * M);
        dma_half(base0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        skp_read_image(img_lane, &E[0]);
        dma_half(base0, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        skp_read_image(img_lane, &E[8]);
    }
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < A.B;
        const char* nbase = reinterpret_cast<const char*>(scores + (size_t)(has_next ? next : tile) * N * M);     // wave-uniform
        // once the row maxima have consumed the raw scores every value of this tile is in registers (and the image was emptied
        // before the compute phase began): the next tile's first half starts to arrive
        sk_fast_tile<NW, CPL>(E, alpha, A.iters, s_colpart, s_b, s_dust, a4, bj, [&] { if (has_next) dma_half(nbase, 0); });
        char* obase = reinterpret_cast<char*>(reinterpret_cast<float*>(A.out) + (size_t)tile * N * M);           // wave-uniform
        auto store_rows = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
            for (int r = r0; r < r0 + 8; ++r) {
                const sk_f2 ar = sk_splat(bcast_lane(a4[r & 3], 16 * (r >> 2)) * tot);
                const sk_f2 lo = E[r][0] * ar * bj[0], hi = E[r][1] * ar * bj[1];
                const sk_u4 o = {__float_as_uint(lo.x), __float_as_uint(lo.y), __float_as_uint(hi.x), __float_as_uint(hi.y)};
                const unsigned vo = voff + (unsigned)(r >> 2) * 4096;
                if ((r & 3) == 0) SK_STORE(vo, o, obase, 0);
                else if ((r & 3) == 1) SK_STORE(vo, o, obase, 1024);
                else if ((r & 3) == 2) SK_STORE(vo, o, obase, 2048);
                else SK_STORE(vo, o, obase, 3072);
            }
        };
        store_rows(0);
        if (has_next) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // everything older than the 8 stores above: the first-half DMA has landed
            skp_read_image(img_lane, &E[0]);
            dma_half(nbase, 1);
        }
        store_rows(8);
        if (!has_next) break;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // second-half DMA (and the first stores) landed; the last 8 stores still drain
        skp_read_image(img_lane, &E[8]);
        tile = next;     // (no barrier: s_colpart / s_b are rewritten only behind the next tile's first barrier)
    }
}

// ---------------------------------------------------------------------------------------------
// streaming kernel: E in a global workspace, arbitrary N, M; compute type T
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* s_scr, int nw) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) s_scr[wave_id()] = v;
    __syncthreads();
    T r = 0;
    for (int k = 0; k < nw; ++k) r += s_scr[k];
    return r;
}

template <typename TIn, typename T, typename TOut>
__global__ __launch_bounds__(1024) void sk_stream_kernel(SkArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = A.N, M = A.M;
    T* s_a = reinterpret_cast<T*>(smem);   // [N+1]
    T* s_b = s_a + (N + 1);                // [M+1]
    T* s_ed = s_b + (M + 1);               // [N]
    T* s_rho = s_ed + N;                   // [N]
    T* s_part = s_rho + N;                 // [4*256]
    T* s_scr = s_part + 1024;              // [32]
    __shared__ int s_cnt[2];

    const int tile = blockIdx.x, t = threadIdx.x, lane = lane_id(), w = wave_id();
    const int nthr = blockDim.x, nw = nthr >> 6;
    const TIn* src = reinterpret_cast<const TIn*>(A.scores) + (size_t)tile * N * M;
    T* Ew = reinterpret_cast<T*>(A.ws) + (size_t)tile * N * M;
    const uint8_t* sm = A.src_mask ? A.src_mask + (size_t)tile * N : nullptr;
    const uint8_t* tm = A.tgt_mask ? A.tgt_mask + (size_t)tile * M : nullptr;
    const bool apply = (A.flags & DR_SK_APPLY_MASK) != 0;
    const bool ragged = (A.flags & DR_SK_RAGGED) != 0;
    const T alpha = (T)(*A.bin_score);

    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        for (int i = t; i < N; i += nthr) c0 += sm ? (sm[i] != 0) : 1;
        for (int j = t; j < M; j += nthr) c1 += tm ? (tm[j] != 0) : 1;
        if (c0) atomicAdd(&s_cnt[0], c0);
        if (c1) atomicAdd(&s_cnt[1], c1);
    }
    __syncthreads();
    const int ms = s_cnt[0], ns = s_cnt[1];
    // marginals are float32 numbers in the reference even for a float64 state (quirk Q22)
    const float normf = -logf((float)(ms + ns));
    const float lmuN = logf((float)ns) + normf, lnuM = logf((float)ms) + normf;
    const T mu = t_exp<T>((T)normf), muN = t_exp<T>((T)lmuN), nu = mu, nuM = t_exp<T>((T)lnuM);

    T xmin = A.shift ? (T)A.shift[tile] : (T)0;
    if (A.flags & DR_SK_MINSHIFT) {
        T mn = (T)INFINITY;
        for (size_t e = t; e < (size_t)N * M; e += nthr) {
            T v = (T)src[e];
            mn = v < mn ? v : mn;
        }
        mn = wave_min(mn);
        if (lane == 0) s_scr[w] = mn;
        __syncthreads();
        mn = s_scr[0];
        for (int k = 1; k < nw; ++k) mn = s_scr[k] < mn ? s_scr[k] : mn;
        xmin = mn;
        __syncthreads();
    }
    auto zval = [&](int i, int j) -> T {
        T v = (T)src[(size_t)i * M + j] - xmin;
        if (apply && ((sm && !sm[i]) || (tm && !tm[j]))) v = -(T)INFINITY;
        return v;
    };

    // pass 0: rho_i, E_ij, first row scaling (b = 1)
    for (int i = w; i < N; i += nw) {
        T m = alpha;
        for (int j = lane; j < M; j += WAVE) {
            T v = zval(i, j);
            m = v > m ? v : m;
        }
        m = wave_max(m);
        T rs = 0;
        for (int j = lane; j < M; j += WAVE) {
            T e = t_exp<T>(zval(i, j) - m);
            Ew[(size_t)i * M + j] = e;
            rs += e;
        }
        rs = wave_sum(rs);
        if (lane == 0) {
            T ed = t_exp<T>(alpha - m);
            s_rho[i] = m;
            s_ed[i] = ed;
            s_a[i] = (ragged && sm && !sm[i]) ? (T)0 : mu / (rs + ed);
        }
    }
    if (t == 0) s_a[N] = muN / ((ragged ? (T)ns : (T)M) + (T)1);
    for (int j = t; j <= M; j += nthr) s_b[j] = (ragged && j < M && tm && !tm[j]) ? (T)0 : (T)1;
    __syncthreads();

    for (int it = 0; it < A.iters; ++it) {
        if (it > 0) {
            // row pass with the current b
            T bs = 0;
            for (int j = t; j < M; j += nthr) bs += s_b[j];
            bs = block_sum(bs, s_scr, nw);
            const T bM = s_b[M];
            __syncthreads();
            for (int i = w; i < N; i += nw) {
                T rs = 0;
                for (int j = lane; j < M; j += WAVE) rs += Ew[(size_t)i * M + j] * s_b[j];
                rs = wave_sum(rs);
                if (lane == 0) s_a[i] = (ragged && sm && !sm[i]) ? (T)0 : mu / (rs + s_ed[i] * bM);
            }
            if (t == 0) s_a[N] = muN / (bs + bM);
            __syncthreads();
        }
        // column pass with the new a
        const T aN = s_a[N];
        {
            T dp = 0;
            for (int i = t; i < N; i += nthr) dp += s_ed[i] * s_a[i];
            dp = block_sum(dp, s_scr, nw);
            if (t == 0) s_b[M] = nuM / (dp + aN);
        }
        const int g = t >> 8, jj = t & 255;     // 4 row groups x 256 columns (blockDim = 1024)
        for (int base = 0; base < M; base += 256) {
            const int j = base + jj;
            T acc = 0;
            if (j < M)
                for (int i = g; i < N; i += 4) acc += Ew[(size_t)i * M + j] * s_a[i];
            s_part[g * 256 + jj] = acc;
            __syncthreads();
            if (t < 256 && j < M)
                s_b[j] = (ragged && tm && !tm[j]) ? (T)0 : nu / (s_part[jj] + s_part[256 + jj] + s_part[512 + jj] + s_part[768 + jj] + aN);
            __syncthreads();
        }
    }

    if (A.flags & DR_SK_OUT_LOG) {
        // log Z = Z + u + v - norm with u_i = log a_i - rho_i, v_j = log b_j (matching.py:89-91)
        TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * (N + 1) * (M + 1);
        const T nrm = (T)normf;
        for (size_t e = t; e < (size_t)(N + 1) * (M + 1); e += nthr) {
            const int i = (int)(e / (M + 1)), j = (int)(e % (M + 1));
            const T z = (i < N && j < M) ? zval(i, j) : alpha;
            const T u = t_log<T>(s_a[i]) - (i < N ? s_rho[i] : alpha);
            const T v = t_log<T>(s_b[j]);
            dst[e] = (TOut)(z + u + v - nrm);
        }
    } else {
        TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * N * M;
        const T S = t_exp<T>(-(T)normf);
        for (size_t e = t; e < (size_t)N * M; e += nthr) {
            const int i = (int)(e / M), j = (int)(e % M);
            dst[e] = (TOut)(Ew[e] * (s_a[i] * S) * s_b[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-workgroup form of the stream kernel for tiles beyond the register-resident path (4DMatch 512 x 512,
// 2D-3D 1024 x 2048: one workgroup per tile left 255 CUs idle and took 4.9 ms per call at 1024 x 2048).
// A tile is cut into G row blocks, one 256-thread workgroup each; the same scaling iteration, split at the
// points where a column sum needs every row block:
//   phase 0   rho_i, E_ij = exp(z_ij - rho_i) -> workspace, a_i from b = 1, column partials sum_i E_ij a_i of the block
//   phase 1   (iters - 1 times) b_j from the partials of all blocks; a_i = mu / (sum_j E_ij b_j + ..); new partials
//   phase 2   b_j from the last partials; out_ij = E_ij a_i b_j e^-norm  (or the log form)
// A wave keeps the row it works on in registers between the row sum and the column accumulation (one read of E per
// phase), its column accumulators too (lane l owns columns l, l + 64, ..): CPL = columns per lane, M <= 64 CPL.
// Launch boundaries are the grid-wide synchronisation; per-tile vectors (a, e^(alpha - rho), rho, the G x M partials,
// the dustbin scalings) live in the workspace behind E.  Deterministic: no atomics, fixed summation order.
// ---------------------------------------------------------------------------------------------
struct SkGridArgs {
    SkArgs k;
    int G, R, phase, it;       // row blocks per tile, rows per block, phase, iteration index of phase 1
    size_t tile_stride;        // workspace elements (of T) per tile
};

template <typename T>
__host__ __device__ inline size_t sk_grid_tile_elems(int N, int M, int G, int iters) {
    return ((size_t)N * M + 3 * (size_t)N + 1 + (size_t)G * M + G + iters + 2 + (size_t)M + 1 + 8 + 3) & ~(size_t)3;   // (tiles start 16-byte aligned)
}

// column sums over the row blocks, between two phases: cb[j] = sum_g cpart[g][j], cb[M] = sum_g dpart[g]
template <typename T>
__global__ __launch_bounds__(256) void sk_grid_reduce_kernel(SkGridArgs GA) {
    const SkArgs& A = GA.k;
    const int N = A.N, M = A.M, G = GA.G, tile = blockIdx.y;
    T* base = reinterpret_cast<T*>(A.ws) + (size_t)tile * GA.tile_stride;
    const T* g_cpart = base + (size_t)N * M + 3 * (size_t)N + 1;
    const T* g_dpart = g_cpart + (size_t)G * M;
    T* g_cb = const_cast<T*>(g_dpart) + G + (A.iters + 2);       // [M + 1]
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < M) {
        T s = 0;
        for (int q = 0; q < G; ++q) s += g_cpart[(size_t)q * M + j];
        g_cb[j] = s;
    } else if (j == M) {
        T s = 0;
        for (int q = 0; q < G; ++q) s += g_dpart[q];
        g_cb[M] = s;
    }
}

// four consecutive elements as one or two 16-byte accesses (the vector form of the grid kernel: M % 4 == 0, 16-byte aligned tiles)
template <typename TS, typename TD> __device__ __forceinline__ void sk_ld4v(const TS* p, TD (&v)[4]) {
    if constexpr (sizeof(TS) == 4) { const float4 x = *reinterpret_cast<const float4*>(p); v[0] = (TD)x.x; v[1] = (TD)x.y; v[2] = (TD)x.z; v[3] = (TD)x.w; }
    else { const double2 x = *reinterpret_cast<const double2*>(p), y = *reinterpret_cast<const double2*>(p + 2); v[0] = (TD)x.x; v[1] = (TD)x.y; v[2] = (TD)y.x; v[3] = (TD)y.y; }
}
template <typename TD, typename TS> __device__ __forceinline__ void sk_st4v(TD* p, const TS (&v)[4]) {
    if constexpr (sizeof(TD) == 4) *reinterpret_cast<float4*>(p) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    else { *reinterpret_cast<double2*>(p) = make_double2((double)v[0], (double)v[1]); *reinterpret_cast<double2*>(p + 2) = make_double2((double)v[2], (double)v[3]); }
}

// VEC: lane l owns columns 4 (l + 64 q) + c (q < CPL / 4, c < 4) and moves them as 16-byte accesses -- 1 KB contiguous per wave
// instruction instead of 256 B (batches of large tiles, e.g. 8 x 1024 x 2048 of the batched 2D-3D loop, are HBM-bound passes);
// the scalar form (lane l owns columns l + 64 k) serves M % 4 != 0 and unaligned tiles.  The column a register holds differs, the
// arithmetic per column and every summation order over ROWS do not; the row sums associate differently, so the last bits may.
template <typename TIn, typename T, typename TOut, int CPL, bool VEC>
__global__ __launch_bounds__(256) void sk_grid_kernel(SkGridArgs GA) {
    const SkArgs& A = GA.k;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = A.N, M = A.M, G = GA.G;
    T* s_b = reinterpret_cast<T*>(smem);         // [M + 1]
    T* s_col = s_b + (M + 1);                    // [4][M] per-wave column partials
    T* s_scr = s_col + 4 * (size_t)M;            // [16]
    __shared__ int s_cnt[2];

    const int tile = blockIdx.y, g = blockIdx.x, t = threadIdx.x, lane = lane_id(), w = wave_id();
    const TIn* src = reinterpret_cast<const TIn*>(A.scores) + (size_t)tile * N * M;
    T* base = reinterpret_cast<T*>(A.ws) + (size_t)tile * GA.tile_stride;
    T* Ew = base;
    T* g_a = Ew + (size_t)N * M;                 // [N + 1]
    T* g_ed = g_a + (N + 1);                     // [N]
    T* g_rho = g_ed + N;                         // [N]
    T* g_cpart = g_rho + N;                      // [G][M]
    T* g_dpart = g_cpart + (size_t)G * M;        // [G]
    T* g_aN = g_dpart + G;                       // [iters + 2]: dustbin-row scaling after pass k
    const T* g_cb = g_aN + (A.iters + 2);        // [M + 1]: column sums over the blocks (sk_grid_reduce_kernel)
    const uint8_t* sm = A.src_mask ? A.src_mask + (size_t)tile * N : nullptr;
    const uint8_t* tm = A.tgt_mask ? A.tgt_mask + (size_t)tile * M : nullptr;
    const bool apply = (A.flags & DR_SK_APPLY_MASK) != 0;
    const bool ragged = (A.flags & DR_SK_RAGGED) != 0;
    const T alpha = (T)(*A.bin_score);

    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        for (int i = t; i < N; i += 256) c0 += sm ? (sm[i] != 0) : 1;
        for (int j = t; j < M; j += 256) c1 += tm ? (tm[j] != 0) : 1;
        if (c0) atomicAdd(&s_cnt[0], c0);
        if (c1) atomicAdd(&s_cnt[1], c1);
    }
    __syncthreads();
    const int ms = s_cnt[0], ns = s_cnt[1];
    const float normf = -logf((float)(ms + ns));                 // float32 marginals (quirk Q22)
    const float lmuN = logf((float)ns) + normf, lnuM = logf((float)ms) + normf;
    const T mu = t_exp<T>((T)normf), muN = t_exp<T>((T)lmuN), nu = mu, nuM = t_exp<T>((T)lnuM);
    const T xmin = A.shift ? (T)A.shift[tile] : (T)0;
    const int r0 = g * GA.R, r1 = min(N, r0 + GA.R);
    auto colj = [&](int k) { return VEC ? 4 * (lane + WAVE * (k >> 2)) + (k & 3) : lane + WAVE * k; };

    // ---- b of the previous pass from the column partials of every block (phases 1, 2) ---------------------
    T aN = muN / ((ragged ? (T)ns : (T)M) + (T)1), bM = 1;
    if (GA.phase > 0) {
        aN = g_aN[GA.it - 1 + 0];                               // a_N the partials were built with
        bM = nuM / (g_cb[M] + aN);
        for (int j = t; j < M; j += 256) s_b[j] = (ragged && tm && !tm[j]) ? (T)0 : nu / (g_cb[j] + aN);
        if (t == 0) s_b[M] = bM;
        __syncthreads();
    }

    if (GA.phase == 2) {
        if (A.flags & DR_SK_OUT_LOG) {
            TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * (N + 1) * (M + 1);
            const T nrm = (T)normf;
            const int re = (g == G - 1) ? N + 1 : r1;           // the last block also writes the dustbin row
            for (int i = r0 + w; i < re; i += 4) {
                const T ai = (i < N) ? g_a[i] : g_aN[GA.it - 1];
                const T u = t_log<T>(ai) - (i < N ? g_rho[i] : alpha);
                for (int j = lane; j <= M; j += WAVE) {
                    T z = alpha;
                    if (i < N && j < M) {
                        z = (T)src[(size_t)i * M + j] - xmin;
                        if (apply && ((sm && !sm[i]) || (tm && !tm[j]))) z = -(T)INFINITY;
                    }
                    dst[(size_t)i * (M + 1) + j] = (TOut)(z + u + t_log<T>(s_b[j]) - nrm);
                }
            }
        } else {
            TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * N * M;
            const T S = t_exp<T>(-(T)normf);
            for (int i = r0 + w; i < r1; i += 4) {
                const T ai = g_a[i] * S;
                if (VEC) {
                    for (int jb = 4 * lane; jb < M; jb += 4 * WAVE) {
                        T e4[4], o4[4];
                        sk_ld4v(Ew + (size_t)i * M + jb, e4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o4[c] = e4[c] * ai * s_b[jb + c];
                        sk_st4v(dst + (size_t)i * M + jb, o4);
                    }
                } else {
                    for (int j = lane; j < M; j += WAVE) dst[(size_t)i * M + j] = (TOut)(Ew[(size_t)i * M + j] * ai * s_b[j]);
                }
            }
        }
        return;
    }

    // ---- phases 0 / 1: rows of this block, one wave per row; the row stays in registers -----------------------
    T bs = 0;
    if (GA.phase == 1) {
        for (int j = t; j < M; j += 256) bs += s_b[j];
        bs = wave_sum(bs);
        if (lane == 0) s_scr[w] = bs;
        __syncthreads();
        bs = s_scr[0] + s_scr[1] + s_scr[2] + s_scr[3];
    }
    const T aN_new = (GA.phase == 0) ? aN : muN / (bs + bM);
    T cacc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) cacc[k] = 0;
    T dacc = 0;
    for (int i = r0 + w; i < r1; i += 4) {
        T e[CPL];
        T ai, ed;
        if (GA.phase == 0) {
            T m = alpha;
            if (VEC) {
#pragma unroll
                for (int q = 0; q < CPL / 4; ++q) {
                    const int jb = 4 * (lane + WAVE * q);
                    T raw[4] = {0, 0, 0, 0};
                    if (jb < M) sk_ld4v(src + (size_t)i * M + jb, raw);
#pragma unroll
                    for (int c = 0; c < 4; ++c) e[4 * q + c] = raw[c];
                }
            }
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int j = colj(k);
                T v = -(T)INFINITY;
                if (j < M) {
                    v = (VEC ? e[k] : (T)src[(size_t)i * M + j]) - xmin;
                    if (apply && ((sm && !sm[i]) || (tm && !tm[j]))) v = -(T)INFINITY;
                }
                e[k] = v;
                m = v > m ? v : m;
            }
            m = wave_max(m);
            T rs = 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int j = colj(k);
                const T ex = (j < M) ? t_exp<T>(e[k] - m) : (T)0;
                e[k] = ex;
                if (!VEC && j < M) Ew[(size_t)i * M + j] = ex;
                rs += ex;
            }
            if (VEC) {
#pragma unroll
                for (int q = 0; q < CPL / 4; ++q) {
                    const int jb = 4 * (lane + WAVE * q);
                    const T e4[4] = {e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
                    if (jb < M) sk_st4v(Ew + (size_t)i * M + jb, e4);
                }
            }
            rs = wave_sum(rs);
            ed = t_exp<T>(alpha - m);
            ai = (ragged && sm && !sm[i]) ? (T)0 : mu / (rs + ed);
            if (lane == 0) { g_rho[i] = m; g_ed[i] = ed; g_a[i] = ai; }
        } else {
            T rs = 0;
            if (VEC) {
#pragma unroll
                for (int q = 0; q < CPL / 4; ++q) {
                    const int jb = 4 * (lane + WAVE * q);
                    T e4[4] = {0, 0, 0, 0};
                    if (jb < M) sk_ld4v(Ew + (size_t)i * M + jb, e4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { e[4 * q + c] = e4[c]; rs += (jb < M) ? e4[c] * s_b[jb + c] : (T)0; }
                }
            } else {
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const int j = lane + WAVE * k;
                    const T ex = (j < M) ? Ew[(size_t)i * M + j] : (T)0;
                    e[k] = ex;
                    rs += (j < M) ? ex * s_b[j] : (T)0;
                }
            }
            rs = wave_sum(rs);
            ed = g_ed[i];
            ai = (ragged && sm && !sm[i]) ? (T)0 : mu / (rs + ed * bM);
            if (lane == 0) g_a[i] = ai;
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) cacc[k] += e[k] * ai;
        dacc += ed * ai;                                        // same value in every lane
    }
    // column partials of the block: 4 waves -> LDS -> workspace
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = colj(k);
        if (j < M) s_col[(size_t)w * M + j] = cacc[k];
    }
    if (lane == 0) s_scr[8 + w] = dacc;
    __syncthreads();
    for (int j = t; j < M; j += 256)
        g_cpart[(size_t)g * M + j] = s_col[j] + s_col[(size_t)M + j] + s_col[2 * (size_t)M + j] + s_col[3 * (size_t)M + j];
    if (t == 0) {
        g_dpart[g] = s_scr[8] + s_scr[9] + s_scr[10] + s_scr[11];
        if (g == 0) g_aN[GA.it] = aN_new;                       // phase 0: it = 0; phase 1: it = 1 .. iters - 1
        if (g == 0 && GA.phase == 0) g_a[N] = aN_new;
    }
}

template <typename TIn, typename T, typename TOut, int CPL>
static int launch_grid_cpl(const SkArgs& a, int G, hipStream_t st) {
    SkGridArgs ga;
    ga.k = a; ga.G = G; ga.R = (a.N + G - 1) / G;
    ga.tile_stride = sk_grid_tile_elems<T>(a.N, a.M, G, a.iters);
    const size_t lds = ((size_t)5 * a.M + 1 + 16) * sizeof(T) + 16;
    // 16-byte accesses when every row of every tile starts 16-byte aligned in the scores, the workspace copy and the output
    const bool log_out = (a.flags & DR_SK_OUT_LOG) != 0;
    const bool vec = a.M % 4 == 0 && !log_out && ((uintptr_t)a.scores % (4 * sizeof(TIn))) == 0 && ((uintptr_t)a.out % (4 * sizeof(TOut))) == 0 &&
                     ((uintptr_t)a.ws % 32) == 0 && ga.tile_stride % 4 == 0 && env_knob("DR_SK_GRID_VEC", 1);
    if (lds > 64 * 1024) {
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)sk_grid_kernel<TIn, T, TOut, CPL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)sk_grid_kernel<TIn, T, TOut, CPL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    for (int ph = 0; ph <= a.iters; ++ph) {
        ga.phase = ph == 0 ? 0 : (ph == a.iters ? 2 : 1);
        ga.it = ph;                                              // phase 1/2 read g_aN[it - 1], phases 0/1 write g_aN[it]
        if (ph > 0) {
            hipLaunchKernelGGL((sk_grid_reduce_kernel<T>), dim3((a.M + 1 + 255) / 256, a.B), dim3(256), 0, st, ga);
            DR_LAUNCH_CHECK();
        }
        if (vec) hipLaunchKernelGGL((sk_grid_kernel<TIn, T, TOut, CPL, true>), dim3(G, a.B), dim3(256), lds, st, ga);
        else hipLaunchKernelGGL((sk_grid_kernel<TIn, T, TOut, CPL, false>), dim3(G, a.B), dim3(256), lds, st, ga);
        DR_LAUNCH_CHECK();
    }
    return DR_OK;
}

// ---------------------------------------------------------------------------------------------
// Co-resident form for tiles beyond the register-resident path (4DMatch 512 x 512 x 8, 2D-3D 1024 x 2048, real 3DMatch pairs of
// 500 - 700 superpoints): ONE launch instead of 2 iters + 1.  A wave owns one row of a tile and keeps its exponentials in
// registers from the load to the store (E never touches memory); a workgroup = 4 rows; the only thing that crosses workgroups is
// the column sums, once per iteration, through global memory in two hops:
//   partials  P[g][0..M] of the workgroup's 4 rows (+ the dustbin column)            -> arrive on counter A
//   slices    workgroup g sums the G partials of ITS slice of columns (fixed order)   -> arrive on counter B
//   everyone reads the M + 1 column sums.
// No atomics on data (bit-reproducible); the exchanged floats are written and read past the L1 and the XCD's L2 (sc1), arrival
// is one sc1 flag word per workgroup polled by the first wave of every workgroup, every spin is bounded.  All workgroups of the
// launch must be resident: 512-thread workgroups (8 rows, one wave each) with <= 128 VGPRs and <= 66 KB of LDS fit 2 per CU; the
// launcher asks the occupancy API for the instantiation it is about to launch (coop_blocks_per_cu) and takes this form only while
// B G <= blocks per CU x CUs / SK_COOP_SHARE, so that SK_COOP_SHARE concurrent launches of it (the engine's streams) still fit
// the chip together -- otherwise the multi-launch grid form.
// A workgroup whose poll gives up (something else holds the CUs: a foreign kernel, a CU mask, a third stream) does NOT carry on
// silently: it sets the sticky device flag g_sk_status (dr_device_status -> DR_ETIMEOUT) and writes NaN to every output entry it
// owns, so the failure is in the data as well.
// ---------------------------------------------------------------------------------------------
constexpr int SK_COOP_SHARE = 2;
constexpr unsigned SK_COOP_SPIN = 1u << 22;
__device__ unsigned g_sk_status;                                   // sticky: bit 0 = a co-resident Sinkhorn timed out
static unsigned g_sk_spin_limit = SK_COOP_SPIN;                    // (dr_debug_sinkhorn_spin_limit: the timeout test)

// arrival flags instead of a counter: a workgroup announces "my stores of pass p have left" by ONE sc1 store of p to its own word
// (no read-modify-write: 256 agent-scope adds to one address serialise at the memory side, ~12 us per hop measured), and the first
// wave of every workgroup polls the whole flag array, 64 words per instruction, until every word has reached p.  Bounded.
// (wave-uniform) false = gave up after `limit` polls: the sticky flag and the tile's status word are set, *s_bad (LDS) tells the workgroup
__device__ __forceinline__ bool sk_wait_flags(const unsigned* flags, int G, unsigned target, unsigned limit, int* status, int* s_bad,
                                              unsigned* call_status) {
    const int lane = threadIdx.x & 63;
    unsigned spins = 0;
    while (true) {
        bool ok = true;
        for (int q = lane; q < G; q += 64) ok = ok && __hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > limit) {
            if (lane == 0) {
                *status = 1; *s_bad = 1; atomicOr(&g_sk_status, 1u);
                if (call_status) atomicOr(call_status, 1u);
            }
            return false;
        }
    }
}
// exchanged data moves as 16-byte accesses that bypass the L1 and the writer's L2 (sc1): two loads per round trip
__device__ __forceinline__ void sk_ld2_sc1(const float* p0, const float* p1, float4& a, float4& b) {
    sk_v4f x, y;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x), "=&v"(y) : "v"(p0), "v"(p1) : "memory");
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w);
}
__device__ __forceinline__ void sk_st_sc1(float* p, const float4& v) {      // (s_nop: VMEM store data hazard, see SK_STORE)
    const sk_v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}

// workspace of the co-resident form per tile: P [G][M4] | cb [M4] | flags A [G64] | flags B [G64] | status      (M4 = M + 1 rounded up to 4)
__host__ __device__ inline int sk_coop_m4(int M) { return (M + 1 + 3) & ~3; }
constexpr int SK_COOP_RW = 8;                                     // rows (waves) per workgroup
__host__ __device__ inline int sk_coop_g(int N) { return (N + SK_COOP_RW - 1) / SK_COOP_RW; }
__host__ __device__ inline int sk_coop_g64(int N) { return (sk_coop_g(N) + 63) & ~63; }
__host__ __device__ inline size_t sk_coop_tile_floats(int N, int M) { return (size_t)(sk_coop_g(N) + 1) * sk_coop_m4(M) + 2 * (size_t)sk_coop_g64(N) + 64; }

template <typename T> struct SkLd4;
template <> struct SkLd4<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) { const float4 x = *reinterpret_cast<const float4*>(p); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
};
template <> struct SkLd4<double> {
    static __device__ __forceinline__ void ld(const double* p, double (&v)[4]) {
        const double2 x = *reinterpret_cast<const double2*>(p), y = *reinterpret_cast<const double2*>(p + 2); v[0] = x.x; v[1] = x.y; v[2] = y.x; v[3] = y.y;
    }
};
template <typename T> struct SkSt4;
template <> struct SkSt4<float> {
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct SkSt4<double> {
    static __device__ __forceinline__ void st(double* p, const float (&v)[4]) {
        *reinterpret_cast<double2*>(p) = make_double2((double)v[0], (double)v[1]); *reinterpret_cast<double2*>(p + 2) = make_double2((double)v[2], (double)v[3]);
    }
};

// RPW = rows per wave: 1, or 2 for batches whose tiles would not all be resident with one row per wave (a wave then adds its rows'
// column partials before they meet the other waves': the sums associate differently, so RPW is part of the result's last bits)
template <typename TIn, typename TOut, int VPL, int RW, int RPW>
__global__ __launch_bounds__(64 * RW) void sk_coop_kernel(SkArgs A) {
    constexpr int NT = 64 * RW;                                   // RW rows (waves) per workgroup
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    constexpr float LOG2E = 1.4426950408889634f;
    const int N = A.N, M = A.M, G = (N + RW * RPW - 1) / (RW * RPW), Mp = VPL * 256, M4 = sk_coop_m4(M), NG4 = M4 / 4;
    float* s_col = reinterpret_cast<float*>(sk_smem);             // [RW][Mp + 4] column partials of the waves; later the M4 column sums
    float* s_red = s_col + RW * (Mp + 4);                         // [RW][8]
    __shared__ int s_cnt[2];
    __shared__ int s_bad;                                         // a poll of this workgroup gave up: its outputs become NaN
    // (tile-major workgroup ids are part of the protocol's progress guarantee: a tile needs only ITS workgroups co-resident, ids are dispatched in
    //  order, so under contention -- a second co-resident launch on another stream -- the lowest unfinished tile always gets the CUs that free up.
    //  Round 5 tried (tile, g) = (id % B, id / B) to put a tile's workgroups on one XCD: 0-3 % faster alone, and two concurrent 8-tile launches
    //  then each held half of EVERY tile and spun into their time-outs: 3-16 s per call instead of 42 ms.  Removed.)
    const int tile = blockIdx.y, g = blockIdx.x, t = threadIdx.x, lane = lane_id(), w = wave_id();
    float* wsb = reinterpret_cast<float*>(A.ws) + (size_t)tile * sk_coop_tile_floats(N, M);
    float* P = wsb;                                               // [G][M4]
    float* cb = P + (size_t)G * M4;                               // [M4]
    unsigned* flagA = reinterpret_cast<unsigned*>(cb + M4);       // [G64] pass whose partials workgroup q has published
    unsigned* flagB = flagA + sk_coop_g64(N);                     // [G64] ... whose slice of column sums
    int* status = reinterpret_cast<int*>(flagB + sk_coop_g64(N));
    const TIn* src = reinterpret_cast<const TIn*>(A.scores) + (size_t)tile * N * M;
    const uint8_t* sm = A.src_mask ? A.src_mask + (size_t)tile * N : nullptr;
    const uint8_t* tm = A.tgt_mask ? A.tgt_mask + (size_t)tile * M : nullptr;
    const bool apply = (A.flags & DR_SK_APPLY_MASK) != 0, ragged = (A.flags & DR_SK_RAGGED) != 0;
    const float alpha = *A.bin_score;

    // ---- marginals (float32, quirk Q22) from the mask counts
    if (t < 2) s_cnt[t] = 0;
    if (t == 2) s_bad = 0;
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        if (sm) { for (int i = t; i < N; i += NT) c0 += sm[i] != 0; } else if (t == 0) c0 = N;
        if (tm) { for (int j = t; j < M; j += NT) c1 += tm[j] != 0; } else if (t == 0) c1 = M;
        if (c0) atomicAdd(&s_cnt[0], c0);
        if (c1) atomicAdd(&s_cnt[1], c1);
    }
    __syncthreads();
    const int ms = s_cnt[0], ns = s_cnt[1];
    const float normf = -logf((float)(ms + ns));
    const float mu = expf(normf), muN = expf(logf((float)ns) + normf), nu = mu, nuM = expf(logf((float)ms) + normf);
    const double xmin_d = A.shift ? A.shift[tile] : 0.0;

    // ---- the wave's rows: scores -> exponentials in registers.  Lane l owns columns 4 (l + 64 k) + c, k < VPL, c < 4, of each of them.
    const int i0 = RW * RPW * g + RPW * w;
    const bool vec_in = (M & 3) == 0 && ((uintptr_t)src % (4 * sizeof(TIn))) == 0;
    float E[RPW][VPL][4], ed[RPW];
    bool rowok[RPW], a_zero[RPW];
    unsigned colmask = 0;                                         // bit 4 k + c: column exists and is not masked (ragged: b = 0 there)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int i = i0 + r;
        rowok[r] = i < N;
        const bool rowmasked = rowok[r] && sm && !sm[i];
        float m = alpha;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int jb = 4 * (lane + 64 * k);
            TIn raw[4] = {0, 0, 0, 0};
            if (rowok[r] && vec_in && jb < M) SkLd4<TIn>::ld(src + (size_t)i * M + jb, raw);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = jb + c;
                float v = -INFINITY;
                if (j < M && rowok[r]) {
                    const TIn x = vec_in ? raw[c] : src[(size_t)i * M + j];
                    v = (float)((double)x - xmin_d);             // (float inputs: exact; float64 state: shifted in float64 like the other paths)
                    if (apply && (rowmasked || (tm && !tm[j]))) v = -INFINITY;
                }
                if (r == 0 && j < M && !(tm && !tm[j])) colmask |= 1u << (4 * k + c);
                E[r][k][c] = v;
                m = fmaxf(m, v);
            }
        }
        m = wave_max(m);
#pragma unroll
        for (int k = 0; k < VPL; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) E[r][k][c] = __builtin_amdgcn_exp2f((E[r][k][c] - m) * LOG2E);   // exp(-inf) = 0: masked / absent entries
        ed[r] = rowok[r] ? __builtin_amdgcn_exp2f((alpha - m) * LOG2E) : 0.f;   // the row's dustbin-column entry
        a_zero[r] = !rowok[r] || (ragged && rowmasked);
    }

    float b[VPL][4];
#pragma unroll
    for (int k = 0; k < VPL; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) b[k][c] = (!ragged || ((colmask >> (4 * k + c)) & 1)) ? 1.f : 0.f;
    float bM = 1.f, aN = muN / ((ragged ? (float)ns : (float)M) + 1.f), ai[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) ai[r] = 0.f;
    // this workgroup's slice of the column sums: float4 groups [g0, g1)
    const int gpw = (NG4 + G - 1) / G, g0 = min(NG4, g * gpw), g1 = min(NG4, g0 + gpw);
    for (int it = 0; it < A.iters; ++it) {
        // a_i = mu / (sum_j E_ij b_j + ed bM)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) s = fmaf(E[r][k][c], b[k][c], s);
            s = wave_sum(s);
            ai[r] = a_zero[r] ? 0.f : mu / (s + ed[r] * bM);
        }
        // column partials of the workgroup's rows -> P[g].  The dustbin column rides at index M: for M < Mp that index lies INSIDE the float4 group of
        // one of the wave's lanes (whose own value there is 0: column M is no column of E), so that lane carries the dustbin partial in its group --
        // ed and ai are wave-uniform, every lane can form it.  (Until round 5 lane 0 stored it with a separate scalar LDS write behind the groups:
        // two stores to the same address from one wave, in an order the compiler was free to change -- a build whose only difference was an unrelated
        // scalar block in front of the kernel put the scalar store first and every tile with M < 2 048 came out with a zero dustbin column.)
        float dsum = ed[0] * ai[0];
#pragma unroll
        for (int r = 1; r < RPW; ++r) dsum = fmaf(ed[r], ai[r], dsum);
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            float4 cp;
            cp.x = E[0][k][0] * ai[0]; cp.y = E[0][k][1] * ai[0]; cp.z = E[0][k][2] * ai[0]; cp.w = E[0][k][3] * ai[0];
#pragma unroll
            for (int r = 1; r < RPW; ++r) {
                cp.x = fmaf(E[r][k][0], ai[r], cp.x); cp.y = fmaf(E[r][k][1], ai[r], cp.y);
                cp.z = fmaf(E[r][k][2], ai[r], cp.z); cp.w = fmaf(E[r][k][3], ai[r], cp.w);
            }
            const int jb = 4 * (lane + 64 * k);
            if (jb == (M & ~3)) {
                const int c = M & 3;
                if (c == 0) cp.x = dsum; else if (c == 1) cp.y = dsum; else if (c == 2) cp.z = dsum; else cp.w = dsum;
            }
            *reinterpret_cast<float4*>(s_col + w * (Mp + 4) + jb) = cp;
        }
        if (M >= Mp && lane == 0) s_col[w * (Mp + 4) + M] = dsum;   // M = Mp: the index is in the row's pad, behind every lane's groups
        __syncthreads();
        for (int q4 = t; q4 < NG4; q4 += NT) {
            float4 v = *reinterpret_cast<const float4*>(s_col + 4 * q4);
#pragma unroll
            for (int r = 1; r < RW; ++r) {                        // fixed order over the workgroup's rows
                const float4 u = *reinterpret_cast<const float4*>(s_col + r * (Mp + 4) + 4 * q4);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            sk_st_sc1(P + (size_t)g * M4 + 4 * q4, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // every store of the workgroup has left
        if (t == 0) __hip_atomic_store(flagA + g, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) sk_wait_flags(flagA, G, (unsigned)(it + 1), A.spin_limit, status, &s_bad, A.call_status);
        __syncthreads();
        // this workgroup's slice: column sums over the G partials
        if constexpr (RW * RPW >= 32) {
            // few partials per tile (G <= 32: the batch form, a workgroup owns 32 rows): a HALF-WAVE per float4 group, lane = partial,
            // NT / 32 groups per round trip, summed by a fixed half-wave butterfly (bit-reproducible), no workgroup barrier inside
            const int hw = t >> 5, l32 = t & 31;
            for (int gq = g0 + hw; gq < g1; gq += NT / 32) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (l32 < G) {
                    sk_v4f y;
                    const float* pq = P + (size_t)l32 * M4 + 4 * gq;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(y) : "v"(pq) : "memory");
                    x = make_float4(y.x, y.y, y.z, y.w);
                }
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) {
                    x.x += __shfl_xor(x.x, m); x.y += __shfl_xor(x.y, m); x.z += __shfl_xor(x.z, m); x.w += __shfl_xor(x.w, m);
                }
                if (l32 == 0) sk_st_sc1(cb + 4 * gq, x);
            }
        } else
        // thread = partial (G <= 256 per pass), two float4 groups per
        // round trip; lanes are summed by the butterfly, the 4 waves in fixed order: bit-reproducible
        for (int gq = g0; gq < g1; gq += 2) {
            float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
            for (int q = t; q < G; q += NT) {
                float4 y0, y1;
                const float* pq = P + (size_t)q * M4;
                sk_ld2_sc1(pq + 4 * gq, pq + 4 * min(gq + 1, NG4 - 1), y0, y1);
                x0.x += y0.x; x0.y += y0.y; x0.z += y0.z; x0.w += y0.w;
                x1.x += y1.x; x1.y += y1.y; x1.z += y1.z; x1.w += y1.w;
            }
            float r8[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) r8[e] = wave_sum(r8[e]);
            if (lane == 0)
#pragma unroll
                for (int e = 0; e < 8; ++e) s_red[w * 8 + e] = r8[e];
            __syncthreads();
            if (t < 2 && gq + t < g1) {
                float4 v = *reinterpret_cast<const float4*>(s_red + 4 * t);
#pragma unroll
                for (int r = 1; r < RW; ++r) {
                    const float4 u = *reinterpret_cast<const float4*>(s_red + 8 * r + 4 * t);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                sk_st_sc1(cb + 4 * (gq + t), v);
            }
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(flagB + g, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) sk_wait_flags(flagB, G, (unsigned)(it + 1), A.spin_limit, status, &s_bad, A.call_status);
        __syncthreads();
        // the M + 1 column sums -> LDS (two float4 groups per thread per round trip), then b_j = nu / (cb_j + a_N), b_M likewise;
        // a_N of the NEXT pass from the new b
        for (int q4 = t; q4 < NG4; q4 += 2 * NT) {
            float4 y0, y1;
            const int q4b = min(q4 + NT, NG4 - 1);
            sk_ld2_sc1(cb + 4 * q4, cb + 4 * q4b, y0, y1);
            *reinterpret_cast<float4*>(s_col + 4 * q4) = y0;
            if (q4 + NT < NG4) *reinterpret_cast<float4*>(s_col + 4 * q4b) = y1;
        }
        __syncthreads();
        float bs = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int jb = 4 * (lane + 64 * k);
            float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jb < M4) cs = *reinterpret_cast<const float4*>(s_col + jb);
            const float cv[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float bj = 0.f;
                if (jb + c < M && (!ragged || ((colmask >> (4 * k + c)) & 1))) bj = nu / (cv[c] + aN);
                b[k][c] = bj;
                bs += bj;
            }
        }
        bM = nuM / (s_col[M] + aN);
        bs = wave_sum(bs);
        aN = muN / (bs + bM);
        __syncthreads();                                          // s_col is rewritten by the next pass
    }
    // ---- out_ij = E_ij a_i b_j e^-norm
    const bool vec_out = (M & 3) == 0 && ((uintptr_t)A.out % (4 * sizeof(TOut))) == 0;
    const bool bad = s_bad != 0;                                  // (every wait above is followed by a workgroup barrier)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        if (!rowok[r]) continue;
        const float S = bad ? __builtin_nanf("") : expf(-normf) * ai[r];
        TOut* dst = reinterpret_cast<TOut*>(A.out) + (size_t)tile * N * M + (size_t)(i0 + r) * M;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int jb = 4 * (lane + 64 * k);
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = E[r][k][c] * S * b[k][c];
            if (vec_out) { if (jb < M) SkSt4<TOut>::st(dst + jb, o); }
            else
#pragma unroll
                for (int c = 0; c < 4; ++c) if (jb + c < M) dst[jb + c] = (TOut)o[c];
        }
    }
}

// rows per wave of the co-resident form for a batch: 1 if the launch is then resident beside a second one, else 2 (tiles of up to 768
// columns: two rows of 3 float4s per lane are 123-125 registers, inside the 128 of two workgroups per CU; 4 float4s are 139), else 0 = not this form
static size_t coop_lds_bytes(int vpl) { return ((size_t)SK_COOP_RW * (vpl * 256 + 4) + 8 * SK_COOP_RW + 8) * sizeof(float); }
// resident workgroups per CU of the instantiation (VPL, RPW) as the occupancy API reports them for its registers and LDS, the least
// over the four (input, output) type pairs, capped by the 2 that 512-thread workgroups of <= 128 registers allow; 0 = not built
static int coop_blocks_per_cu(int vpl, int rpw) {
    static int cache[9][3];                                       // 0 = not asked yet, -1 = not built / query failed
    if (vpl < 1 || vpl > 8 || rpw < 1 || rpw > 2 || (rpw == 2 && vpl > 3)) return 0;
    int& c = cache[vpl][rpw];
    if (c == 0) {
        int least = 2;
        bool ok = true;
        auto ask = [&](const void* fn) {
            int nb = 0;
            const size_t lds = coop_lds_bytes(vpl);
            if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) ok = false;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * SK_COOP_RW, lds) != hipSuccess) ok = false;
            least = nb < least ? nb : least;
        };
#define SK_COOP_ASK(V, R)                                                         \
    ask((const void*)sk_coop_kernel<float, float, V, SK_COOP_RW, R>);             \
    ask((const void*)sk_coop_kernel<double, float, V, SK_COOP_RW, R>);            \
    ask((const void*)sk_coop_kernel<double, double, V, SK_COOP_RW, R>);
        switch (vpl * 4 + rpw) {
            case 1 * 4 + 1: SK_COOP_ASK(1, 1) break;
            case 1 * 4 + 2: SK_COOP_ASK(1, 2) break;
            case 2 * 4 + 1: SK_COOP_ASK(2, 1) break;
            case 2 * 4 + 2: SK_COOP_ASK(2, 2) break;
            case 3 * 4 + 1: SK_COOP_ASK(3, 1) break;
            case 3 * 4 + 2: SK_COOP_ASK(3, 2) break;
            case 4 * 4 + 1: SK_COOP_ASK(4, 1) break;
            case 5 * 4 + 1: SK_COOP_ASK(5, 1) break;
            case 6 * 4 + 1: SK_COOP_ASK(6, 1) break;
            case 7 * 4 + 1: SK_COOP_ASK(7, 1) break;
            case 8 * 4 + 1: SK_COOP_ASK(8, 1) break;
            default: ok = false;
        }
#undef SK_COOP_ASK
        c = (ok && least > 0) ? least : -1;
    }
    return c > 0 ? c : 0;
}
static int coop_rows_per_wave(int B, int N, int M, int flags) {
    if (flags & (DR_SK_MINSHIFT | DR_SK_STRICT | DR_SK_OUT_LOG)) return 0;
    if (M > 2048 || !env_knob("DR_SK_COOP", 1)) return 0;
    const int n_cu = device_cu_count();
    const int vpl = (M + 255) / 256;
    // residency as the runtime reports it for the instantiation that would run, shared with SK_COOP_SHARE - 1 other launches
    if ((long)B * sk_coop_g(N) <= (long)coop_blocks_per_cu(vpl, 1) * n_cu / SK_COOP_SHARE) return 1;
    if (M <= 768 && env_knob("DR_SK_COOP", 1) != 2 &&
        (long)B * ((N + 2 * SK_COOP_RW - 1) / (2 * SK_COOP_RW)) <= (long)coop_blocks_per_cu(vpl, 2) * n_cu / SK_COOP_SHARE) return 2;
    return 0;
}
// The BATCH form of the co-resident kernel (round 5; cfg5's 8 x 1024 x 2048 per call): the register files of the chip hold the exponentials of
// the whole batch.  A wave keeps EIGHT rows (256 registers of E at 2 048 columns), a workgroup is 4 waves -- one per SIMD, one workgroup per CU
// at ~330 registers -- so 32 rows per CU and 8 192 rows on the chip: 8 tiles of 1 024 rows exactly.  One launch instead of 2 iters + 1, E never
// touches memory, a tile's column sums cross its 32 workgroups once per iteration (two hops).  Residency: the synchronisation domain is ONE
// TILE (its N / 32 workgroups, consecutive block indices), not the launch: a launch that fits the chip alone is always safe, and beside other
// launches (the engine's concurrent calls) every fully resident tile finishes by itself and hands its CUs on, so at most the frontier tile
// of each launch is ever waiting for CUs -- with the in-order dispatch of a queue that cannot deadlock; the spins stay bounded all the same
// (DR_ETIMEOUT instead of a hang if the hardware ever dispatched out of order).
constexpr int SK_BATCH_RW = 4, SK_BATCH_RPW = 8;
static size_t coop_batch_lds_bytes(int vpl) { return ((size_t)SK_BATCH_RW * (vpl * 256 + 4) + 8 * SK_BATCH_RW + 8) * sizeof(float); }
// (two instantiations per type pair: 4 or 8 float4 groups per lane and row.  A 6-group one -- 1 025 .. 1 536 columns -- faulted on the device
//  for M < 1 536 in the first hardware run and was dropped unexplained: such tiles take the 8-group kernel with idle lanes)
static int coop_batch_vpl(int M) { return M <= 1024 ? 4 : 8; }
static bool coop_batch_path(int B, int N, int M, int flags) {
    if (flags & (DR_SK_MINSHIFT | DR_SK_STRICT | DR_SK_OUT_LOG)) return false;
    if (M > 2048 || M <= 768 || !env_knob("DR_SK_COOP", 1) || !env_knob("DR_SK_BATCH", 1)) return false;
    const int n_cu = device_cu_count();
    const long G = (N + SK_BATCH_RW * SK_BATCH_RPW - 1) / (SK_BATCH_RW * SK_BATCH_RPW);
    return G <= 32 && (long)B * G <= n_cu;
}
static bool coop_path(int B, int N, int M, int flags) { return coop_rows_per_wave(B, N, M, flags) != 0 || coop_batch_path(B, N, M, flags); }

// flags + status of every tile start at zero: they sit behind the G partials and the column sums of the launch's G inside each tile's workspace
// slice.  ONE launch for the batch (a hipMemsetAsync per tile was 8 x 4.8 us in front of every Sinkhorn of an 8-pair call: 183 memset nodes, 0.9 ms
// of cfg5's 26 ms graph; 350 nodes in cfg3's)
__global__ __launch_bounds__(256) void sk_coop_zero_kernel(float* __restrict__ base, size_t stride, int n) {
    float* p = base + (size_t)blockIdx.x * stride;
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0.f;
}
static int coop_zero_flags(const SkArgs& a, int G, hipStream_t st) {
    const size_t tf = sk_coop_tile_floats(a.N, a.M);
    hipLaunchKernelGGL(sk_coop_zero_kernel, dim3(a.B), dim3(256), 0, st, reinterpret_cast<float*>(a.ws) + (size_t)(G + 1) * sk_coop_m4(a.M), tf,
                       (int)(2 * (size_t)sk_coop_g64(a.N) + 64));
    DR_LAUNCH_CHECK();
    return DR_OK;
}

template <typename TIn, typename TOut>
static int launch_coop_batch(const SkArgs& a, hipStream_t st) {
    const int G = (a.N + SK_BATCH_RW * SK_BATCH_RPW - 1) / (SK_BATCH_RW * SK_BATCH_RPW), vpl = coop_batch_vpl(a.M);
    // the kernel's half-wave column-sum reduction (RW * RPW >= 32) reads ONE partial per lane of a half-wave: more than 32 partials would be dropped
    static_assert(SK_BATCH_RW * SK_BATCH_RPW >= 32, "the batch form is the instantiation with the half-wave reduction");
    if (G > 32) return DR_EINVAL;
    const int zrc = coop_zero_flags(a, G, st);
    if (zrc) return zrc;
    const dim3 grid(G, a.B), blk(64 * SK_BATCH_RW);
    const size_t lds = coop_batch_lds_bytes(vpl);
#define SK_BATCH_LAUNCH(V)                                                                                                                   \
    {                                                                                                                                        \
        if (lds > 64 * 1024) DR_HIP_CHECK(hipFuncSetAttribute((const void*)sk_coop_kernel<TIn, TOut, V, SK_BATCH_RW, SK_BATCH_RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((sk_coop_kernel<TIn, TOut, V, SK_BATCH_RW, SK_BATCH_RPW>), grid, blk, lds, st, a);                               \
    }
    if (vpl == 4) SK_BATCH_LAUNCH(4) else SK_BATCH_LAUNCH(8)
#undef SK_BATCH_LAUNCH
    DR_LAUNCH_CHECK();
    return DR_OK;
}

template <typename TIn, typename TOut>
static int launch_coop(const SkArgs& a, hipStream_t st) {
    const int rpw = coop_rows_per_wave(a.B, a.N, a.M, a.flags);
    if (!rpw) return DR_EINVAL;
    const int G = (a.N + SK_COOP_RW * rpw - 1) / (SK_COOP_RW * rpw), vpl = (a.M + 255) / 256;
    static_assert(SK_COOP_RW * 2 < 32, "an instantiation of >= 32 rows per workgroup takes the half-wave reduction: it then needs the G <= 32 guard of the batch form");
    const int zrc = coop_zero_flags(a, G, st);
    if (zrc) return zrc;
    const dim3 grid(G, a.B), blk(64 * SK_COOP_RW);
#define SK_COOP_LAUNCH(V, R)                                                                                                 \
    {                                                                                                                        \
        const size_t lds = coop_lds_bytes(V);                                                                                \
        if (lds > 64 * 1024) DR_HIP_CHECK(hipFuncSetAttribute((const void*)sk_coop_kernel<TIn, TOut, V, SK_COOP_RW, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((sk_coop_kernel<TIn, TOut, V, SK_COOP_RW, R>), grid, blk, lds, st, a);                            \
    }
#define SK_COOP_CASE(V) case V: SK_COOP_LAUNCH(V, 1) break;
#define SK_COOP_CASE2(V) case V: if (rpw == 2) SK_COOP_LAUNCH(V, 2) else SK_COOP_LAUNCH(V, 1) break;
    switch (vpl) {
        SK_COOP_CASE2(1) SK_COOP_CASE2(2) SK_COOP_CASE2(3) SK_COOP_CASE(4) SK_COOP_CASE(5) SK_COOP_CASE(6) SK_COOP_CASE(7) SK_COOP_CASE(8)
        default: return DR_ENOSUP;
    }
#undef SK_COOP_CASE
#undef SK_COOP_CASE2
#undef SK_COOP_LAUNCH
    DR_LAUNCH_CHECK();
    return DR_OK;
}

static int sk_grid_blocks(int B, int N) {
    // enough workgroups for the chip, at least 8 rows (two per wave) per block, at most 128 blocks (the column
    // partials are G x M values per tile)
    const int target = env_knob("DR_SK_GRID_WGS", 512), cap = env_knob("DR_SK_GRID_CAP", 128);   // (512 = two workgroups per CU: 8 x 1024 x 2048 takes 203 us against 222 at 768 and 292 at 1536, profiles/r04_sinkhorn_grid_vec.txt)
    int G = (target + B - 1) / B;
    if (G > cap) G = cap;
    if (G > (N + 7) / 8) G = (N + 7) / 8;
    return G < 1 ? 1 : G;
}

static bool grid_path(int N, int M, int flags) { return !(flags & DR_SK_MINSHIFT) && M <= 64 * 32; }

template <typename TIn, typename T, typename TOut>
static int launch_grid(const SkArgs& a, hipStream_t st) {
    const int G = sk_grid_blocks(a.B, a.N);
    if (a.M <= 64 * 8) return launch_grid_cpl<TIn, T, TOut, 8>(a, G, st);
    if (a.M <= 64 * 16) return launch_grid_cpl<TIn, T, TOut, 16>(a, G, st);
    return launch_grid_cpl<TIn, T, TOut, 32>(a, G, st);
}

static size_t stream_lds_bytes(int N, int M, size_t esz) { return ((size_t)3 * N + M + 2 + 1024 + 32) * esz + 16; }

template <typename TIn, typename T, typename TOut>
static int launch_stream(const SkArgs& a, hipStream_t st) {
    size_t lds = stream_lds_bytes(a.N, a.M, sizeof(T));
    if (lds > 160 * 1024) return DR_ENOSUP;
    if (lds > 64 * 1024)
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)sk_stream_kernel<TIn, T, TOut>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((sk_stream_kernel<TIn, T, TOut>), dim3(a.B), dim3(1024), lds, st, a);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

template <typename TIn, typename TOut, bool VEC>
static int launch_reg2(const SkArgs& a, hipStream_t st) {
    if (a.N <= 128 && a.M <= 128) {
        hipLaunchKernelGGL((sk_reg_kernel<TIn, TOut, 8, 2, VEC>), dim3(a.B), dim3(512), 0, st, a);
    } else if (a.N <= 128) {
        hipLaunchKernelGGL((sk_reg_kernel<TIn, TOut, 8, 4, VEC>), dim3(a.B), dim3(512), 0, st, a);
    } else {
        hipLaunchKernelGGL((sk_reg_kernel<TIn, TOut, 16, 4, VEC>), dim3(a.B), dim3(1024), 0, st, a);
    }
    DR_LAUNCH_CHECK();
    return DR_OK;
}
template <typename TIn, typename TOut>
static int launch_reg(const SkArgs& a, hipStream_t st) {
    const bool plain = a.vec_in && a.vec_out && !a.src_mask && !a.tgt_mask && !(a.flags & DR_SK_MINSHIFT);
    if (plain && a.N == 256 && a.M == 256) {
        if (a.B >= SK_NT_MIN_TILES && std::is_same<TIn, float>::value && std::is_same<TOut, float>::value && !a.shift &&
            sk_persist_grid() > 0) {
            static bool attr_done = false;
            if (!attr_done) {
                DR_HIP_CHECK(hipFuncSetAttribute((const void*)sk_fast_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)SKP_LDS));
                attr_done = true;
            }
            const int G = a.B < sk_persist_grid() ? a.B : sk_persist_grid();
            hipLaunchKernelGGL(sk_fast_persist_kernel, dim3(G), dim3(1024), SKP_LDS, st, a);
        } else if (a.B >= SK_NT_MIN_TILES && sizeof(TIn) == 4 && sizeof(TOut) == 4)   // (float64 tiles measured slower with nt)
            hipLaunchKernelGGL((sk_fast_kernel<TIn, TOut, 16, 4, true>), dim3(a.B), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((sk_fast_kernel<TIn, TOut, 16, 4, false>), dim3(a.B), dim3(1024), 0, st, a);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    if (plain && a.N == 128 && a.M == 128) {
        if (a.B >= 4 * SK_NT_MIN_TILES && sizeof(TIn) == 4 && sizeof(TOut) == 4)
            hipLaunchKernelGGL((sk_fast_kernel<TIn, TOut, 8, 2, true>), dim3(a.B), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((sk_fast_kernel<TIn, TOut, 8, 2, false>), dim3(a.B), dim3(512), 0, st, a);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    if (a.vec_in && a.vec_out) return launch_reg2<TIn, TOut, true>(a, st);
    return launch_reg2<TIn, TOut, false>(a, st);
}

static bool reg_path(int N, int M, int flags) {
    return !(flags & (DR_SK_STRICT | DR_SK_OUT_LOG)) && N <= 256 && M <= 256;
}

// workspace of the non-register paths: E per tile (stream kernel), plus the per-tile vectors of the grid form
// (sized for the largest iteration count the library is used with, so that the size does not depend on it)
static size_t sk_workspace_need(int B, int N, int M, int esz, int flags, int iters) {
    if (!grid_path(N, M, flags)) return (size_t)B * N * M * esz;
    // (the co-resident form's exchange area, B sk_coop_tile_floats(N, M) floats ~ a quarter of E, fits inside the grid form's workspace)
    const int G = sk_grid_blocks(B, N);
    return (size_t)B * sk_grid_tile_elems<float>(N, M, G, iters > 16 ? iters : 16) * esz;
}

template <typename TIn>
static int sinkhorn_dispatch(int B, int N, int M, const TIn* scores, const double* shift, const uint8_t* src_mask,
                             const uint8_t* tgt_mask, const float* bin_score, int iters, int flags, void* out, void* ws,
                             size_t ws_bytes, void* stream, unsigned* call_status = nullptr) {
    if (B < 0 || N < 1 || M < 1 || iters < 1 || !scores || !bin_score || !out) return DR_EINVAL;
    if (B == 0) return DR_OK;
    constexpr bool in64 = sizeof(TIn) == 8;
    const bool out32 = !in64 || (flags & DR_SK_OUT_F32);
    SkArgs a;
    a.scores = scores; a.src_mask = src_mask; a.tgt_mask = tgt_mask; a.bin_score = bin_score;
    a.out = out; a.ws = ws; a.shift = shift; a.B = B; a.N = N; a.M = M; a.iters = iters; a.flags = flags;
    a.spin_limit = g_sk_spin_limit; a.call_status = call_status;
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes: read the score tile once, write the conf tile once
    ProfScope ps(PK_SINKHORN, (double)B * N * M * (sizeof(TIn) + (out32 ? 4.0 : 8.0)), st);
    if (reg_path(N, M, flags)) {
        const int cpl = (N <= 128 && M <= 128) ? 2 : 4;
        const size_t in_al = cpl * sizeof(TIn) > 16 ? 16 : cpl * sizeof(TIn);
        const size_t osz = out32 ? 4 : 8;
        const size_t out_al = cpl * osz > 16 ? 16 : cpl * osz;
        a.vec_in = (M % cpl == 0) && ((uintptr_t)scores % in_al == 0);
        a.vec_out = (M % cpl == 0) && ((uintptr_t)out % out_al == 0);
        if (out32) return launch_reg<TIn, float>(a, st);
        return launch_reg<TIn, double>(a, st);
    }
    const bool strict64 = in64 && (flags & DR_SK_STRICT);
    const size_t need = sk_workspace_need(B, N, M, strict64 ? 8 : 4, flags, iters);
    if (!ws || ws_bytes < need) return DR_EWORKSPACE;
    a.vec_in = a.vec_out = 0;
    if (coop_rows_per_wave(B, N, M, flags) != 0) {
        if (out32) return launch_coop<TIn, float>(a, st);
        return launch_coop<TIn, double>(a, st);
    }
    if (coop_batch_path(B, N, M, flags)) {
        if (out32) return launch_coop_batch<TIn, float>(a, st);
        return launch_coop_batch<TIn, double>(a, st);
    }
    if (grid_path(N, M, flags)) {
        if (strict64) {
            if (out32) return launch_grid<TIn, double, float>(a, st);
            return launch_grid<TIn, double, double>(a, st);
        }
        if (out32) return launch_grid<TIn, float, float>(a, st);
        return launch_grid<TIn, float, double>(a, st);
    }
    if (strict64) {
        if (out32) return launch_stream<TIn, double, float>(a, st);
        return launch_stream<TIn, double, double>(a, st);
    }
    if (out32) return launch_stream<TIn, float, float>(a, st);
    return launch_stream<TIn, float, double>(a, st);
}

int sinkhorn_f32(int B, int N, int M, const float* scores, const uint8_t* sm, const uint8_t* tm, const float* bin_score,
                 int iters, int flags, float* out, void* ws, size_t ws_bytes, hipStream_t st, unsigned* call_status) {
    return sinkhorn_dispatch<float>(B, N, M, scores, nullptr, sm, tm, bin_score, iters, flags, out, ws, ws_bytes, st, call_status);
}
int sinkhorn_f64(int B, int N, int M, const double* scores, const double* shift, const uint8_t* sm, const uint8_t* tm,
                 const float* bin_score, int iters, int flags, void* out, void* ws, size_t ws_bytes, hipStream_t st, unsigned* call_status) {
    return sinkhorn_dispatch<double>(B, N, M, scores, shift, sm, tm, bin_score, iters, flags, out, ws, ws_bytes, st, call_status);
}

// fp16 tiles in, fp16 confidences out: the register-resident kernel only (tiles up to 256 x 256: BASELINE cfg1 / cfg2)
int sinkhorn_f16(int B, int N, int M, const void* scores, const uint8_t* sm, const uint8_t* tm, const float* bin_score, int iters, int flags,
                 void* out, hipStream_t st) {
    if (B < 0 || N < 1 || M < 1 || iters < 1 || !scores || !bin_score || !out) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!reg_path(N, M, flags)) return DR_ENOSUP;
    SkArgs a;
    a.scores = scores; a.src_mask = sm; a.tgt_mask = tm; a.bin_score = bin_score;
    a.out = out; a.ws = nullptr; a.shift = nullptr; a.B = B; a.N = N; a.M = M; a.iters = iters; a.flags = flags;
    a.spin_limit = g_sk_spin_limit; a.call_status = nullptr;
    ProfScope ps(PK_SINKHORN, (double)B * N * M * 4.0, st);
    const int cpl = (N <= 128 && M <= 128) ? 2 : 4;
    a.vec_in = (M % cpl == 0) && ((uintptr_t)scores % (2 * cpl) == 0);
    a.vec_out = (M % cpl == 0) && ((uintptr_t)out % (2 * cpl) == 0);
    const bool plain = a.vec_in && a.vec_out && !sm && !tm && !(flags & DR_SK_MINSHIFT);
    if (plain && N == 256 && M == 256) {              // the headline's tile: the per-tile fast kernel (16 waves x 16 rows, 4 columns per lane)
        hipLaunchKernelGGL((sk_fast_kernel<_Float16, _Float16, 16, 4, false>), dim3(B), dim3(1024), 0, st, a);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    if (plain && N == 128 && M == 128) {
        hipLaunchKernelGGL((sk_fast_kernel<_Float16, _Float16, 8, 2, false>), dim3(B), dim3(512), 0, st, a);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    if (a.vec_in && a.vec_out) return launch_reg2<_Float16, _Float16, true>(a, st);
    return launch_reg2<_Float16, _Float16, false>(a, st);
}

// waits for the stream, reads (and clears) the sticky flag
int sinkhorn_device_status(hipStream_t st, bool clear) {
    DR_HIP_CHECK(hipStreamSynchronize(st));
    unsigned h = 0;
    DR_HIP_CHECK(hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_sk_status), sizeof(h)));
    if (h && clear) {
        const unsigned z = 0;
        DR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sk_status), &z, sizeof(z)));
    }
    return (h & 1u) ? DR_ETIMEOUT : DR_OK;
}

// waits for the stream, reads (and clears) a caller's own sticky word (the loops' per-workspace status)
int sinkhorn_call_status(unsigned* word, hipStream_t st, bool clear) {
    DR_HIP_CHECK(hipStreamSynchronize(st));
    unsigned h = 0;
    DR_HIP_CHECK(hipMemcpy(&h, word, sizeof(h), hipMemcpyDeviceToHost));
    if (h && clear) DR_HIP_CHECK(hipMemset(word, 0, sizeof(h)));
    return (h & 3u) ? DR_ETIMEOUT : DR_OK;      // bit 0: a co-resident Sinkhorn; bit 1: the k-split exchange of a LayerNorm GEMM (pgemm.h)
}

}  // namespace dr

extern "C" {

int dr_device_status(void* stream, int clear) { return dr::sinkhorn_device_status((hipStream_t)stream, clear != 0); }
/* diagnostics (include/diffreg_hip_debug.h): polls before the co-resident Sinkhorn gives up; 0 = the default */
void dr_debug_sinkhorn_spin_limit(unsigned polls) { dr::g_sk_spin_limit = polls ? polls : dr::SK_COOP_SPIN; }

size_t dr_sinkhorn_workspace_bytes(int B, int N, int M, int elem_bytes, int flags) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    if (dr::reg_path(N, M, flags)) return 0;
    const bool strict64 = elem_bytes == 8 && (flags & DR_SK_STRICT);
    return dr::sk_workspace_need(B, N, M, strict64 ? 8 : 4, flags, 16);
}

int dr_sinkhorn_f32(int B, int N, int M, const float* scores, const uint8_t* src_mask, const uint8_t* tgt_mask,
                    const float* bin_score, int iters, int flags, float* out, void* workspace, size_t workspace_bytes,
                    void* stream) {
    return dr::sinkhorn_dispatch<float>(B, N, M, scores, nullptr, src_mask, tgt_mask, bin_score, iters, flags, out, workspace,
                                        workspace_bytes, stream);
}

int dr_sinkhorn_f16(int B, int N, int M, const void* scores, const uint8_t* src_mask, const uint8_t* tgt_mask, const float* bin_score,
                    int iters, int flags, void* out, void* stream) {
    return dr::sinkhorn_f16(B, N, M, scores, src_mask, tgt_mask, bin_score, iters, flags, out, (hipStream_t)stream);
}

int dr_sinkhorn_f64(int B, int N, int M, const double* scores, const uint8_t* src_mask, const uint8_t* tgt_mask,
                    const float* bin_score, int iters, int flags, void* out, void* workspace, size_t workspace_bytes,
                    void* stream) {
    return dr::sinkhorn_dispatch<double>(B, N, M, scores, nullptr, src_mask, tgt_mask, bin_score, iters, flags, out, workspace,
                                         workspace_bytes, stream);
}

}  // extern "C"
