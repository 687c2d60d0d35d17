// attention_bwd.hip -- fused backward of softmax(q k^T / sqrt(d)) v per head (3D/models/transformero.py:79-85) for the training path
// (second half): no [B, H, L, S] matrix is ever materialised.  Token layout like the forward kernels: q / o / d o rows
// = B L, k / v rows = B S, head h in columns [h d, (h + 1) d).  Three launches, all on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact
// fp32 products, so the gradients carry fp32 rounding only):
//   stats  per query: lse = log sum_j exp(scale s_j) over the live keys, delta = sum_f dO_f O_f
//   dQ     own = a block of 32 queries, sweep over the key blocks:   S^T = K Q^T -> P^T, dP^T = V dO^T, dS^T = scale P^T (dP^T - delta);  dQ^T += K^T dS^T
//   dK|dV  own = a block of 32 keys,    sweep over the query blocks: S = Q K^T -> P,  dP = dO V^T,   dS = scale P (dP - delta);      dK^T += Q^T dS, dV^T += dO^T P
// One kernel body: lanes = the OWN index (the MFMA's column), registers = the OTHER index (its rows), so P and dS leave the
// score registers straight into the next MFMA as its B operand (k = the other index, in the register order row(e, h) -- the A operand
// reads the other block's LDS tile at the same rows).  The waves of a workgroup take different OTHER blocks and add their accumulators
// through LDS at the end in a fixed order: bit-reproducible, no atomics.  Masks as the training forward applies them (dr_softmax_rows_f32):
// key j is dead for query l when q_mask[l] && !k_mask[j].
#include "kernels.h"
#include <string.h>

namespace dr {
namespace {

typedef float bf32x16 __attribute__((ext_vector_type(16)));

struct AttnBwdArgs {
    const float *q, *k, *v, *o, *go;
    float *dq, *dk, *dv;
    const uint8_t *qmask, *kmask;          // [B L], [B S] or null (both or none)
    float *lse, *delta;                    // [B H L]
    int ldq, ldk, ldv, ldo, ldg, lddq, lddk, lddv;
    int B, H, L, S, d;
    float scale;
};

__device__ __forceinline__ int row_of(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }     // row of accumulator register e in lane half h
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 32 rows x d floats of `base` (row stride ld, rows clamped to [0, nrows)) -> dst[r * TS + f]   (one wave)
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int ld, int row0, int nrows, int d, int TS, float* dst, int lane) {
    const int d4 = d >> 2;
    for (int idx = lane; idx < 32 * d4; idx += 64) {
        const int r = idx / d4, c4 = idx - r * d4;
        const int gr = min(row0 + r, nrows - 1);
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)gr * ld + 4 * c4);
        float* p = dst + r * TS + 4 * c4;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
}

// MODE 0: stats (own = queries);  1: dQ (own = queries, other = keys);  2: dK | dV (own = keys, other = queries)
template <int DT, int NW, int MODE>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(AttnBwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int d = A.d, TS = d + 1, TILE = 32 * TS + 32;              // (+ 32: a transposed read of the last row may run past its end)
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.z, head = blockIdx.y, own0 = blockIdx.x * 32;
    const int L = A.L, S = A.S;
    constexpr bool OWN_Q = MODE != 2;
    const int own_len = OWN_Q ? L : S, oth_len = OWN_Q ? S : L;
    float* own_x = smem;
    float* own_y = own_x + TILE;
    float* wave_base = own_y + TILE;
    float* oth_x = wave_base + (size_t)w * 2 * TILE;
    float* oth_y = oth_x + TILE;
    const size_t hoff = (size_t)head * d;
    const float* Q = A.q + (size_t)b * L * A.ldq + hoff;
    const float* K = A.k + (size_t)b * S * A.ldk + hoff;
    const float* V = A.v + (size_t)b * S * A.ldv + hoff;
    const float* GO = A.go + (size_t)b * L * A.ldg + hoff;
    const float* Xown = OWN_Q ? Q : K; const int ldxo = OWN_Q ? A.ldq : A.ldk;
    const float* Yown = OWN_Q ? GO : V; const int ldyo = OWN_Q ? A.ldg : A.ldv;
    const float* Xoth = OWN_Q ? K : Q; const int ldxt = OWN_Q ? A.ldk : A.ldq;
    const float* Yoth = OWN_Q ? V : GO; const int ldyt = OWN_Q ? A.ldv : A.ldg;
    // ---- the own tiles, loaded by the waves in turn
    if (w == 0) load_tile(Xown, ldxo, own0, own_len, d, TS, own_x, lane);
    if (MODE != 0 && w == (NW > 1 ? 1 : 0)) load_tile(Yown, ldyo, own0, own_len, d, TS, own_y, lane);
    __syncthreads();
    const size_t bh = ((size_t)b * A.H + head);
    const uint8_t* qm = A.qmask ? A.qmask + (size_t)b * L : nullptr;
    const uint8_t* km = A.kmask ? A.kmask + (size_t)b * S : nullptr;
    const int own_i = own0 + l31;
    const bool own_in = own_i < own_len;

    if (MODE == 0) {
        // ---- lse over the live keys: per wave an online (max, sum) over its key blocks, combined through LDS
        const bool qv = qm ? (qm[min(own_i, L - 1)] != 0) : true;
        float m_run = -INFINITY, l_run = 0.f;
        for (int ob = w; ob * 32 < oth_len; ob += NW) {
            load_tile(Xoth, ldxt, ob * 32, oth_len, d, TS, oth_x, lane);
            wave_lds_sync();
            bf32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
            const float* ap = oth_x + l31 * TS + h;
            const float* bp = own_x + l31 * TS + h;
            for (int s2 = 0; s2 < d; s2 += 2) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s2], bp[s2], sc, 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = ob * 32 + row_of(r, h);
                const bool dead = key >= S || (km && qv && !km[min(key, S - 1)]);
                sc[r] = dead ? -INFINITY : sc[r] * A.scale;
                mx = fmaxf(mx, sc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            if (m_new > -INFINITY) {
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += expf(sc[r] - m_new);
                ps += __shfl_xor(ps, 32);
                l_run = l_run * expf(m_run - m_new) + ps;
                m_run = m_new;
            }
            wave_lds_sync();
        }
        float* s_stat = wave_base + (size_t)NW * 2 * TILE;                // [NW][32][2]
        if (h == 0) { s_stat[(w * 32 + l31) * 2] = m_run; s_stat[(w * 32 + l31) * 2 + 1] = l_run; }
        __syncthreads();
        if (w == 0 && h == 0 && own_in) {
            float m = -INFINITY;
            for (int k2 = 0; k2 < NW; ++k2) m = fmaxf(m, s_stat[(k2 * 32 + l31) * 2]);
            float l = 0.f;
            for (int k2 = 0; k2 < NW; ++k2) {
                const float mk = s_stat[(k2 * 32 + l31) * 2];
                if (mk > -INFINITY) l += s_stat[(k2 * 32 + l31) * 2 + 1] * expf(mk - m);
            }
            A.lse[bh * L + own_i] = l > 0.f ? m + logf(l) : INFINITY;        // (no live key: every probability is 0)
        }
        // delta[q] = sum_f dO[q][f] O[q][f]: one wave per query row
        const float* O = A.o + (size_t)b * L * A.ldo + hoff;
        for (int qq = w; qq < 32; qq += NW) {
            const int qi = own0 + qq;
            if (qi >= L) break;
            float s = 0.f;
            for (int f = lane; f < d; f += 64) s = fmaf(GO[(size_t)qi * A.ldg + f], O[(size_t)qi * A.ldo + f], s);
            s = wave_sum(s);
            if (lane == 0) A.delta[bh * L + qi] = s;
        }
        return;
    }

    // ---- gradient sweeps
    bf32x16 g1[DT], g2[MODE == 2 ? DT : 1];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { g1[i][r] = 0.f; if (MODE == 2) g2[i][r] = 0.f; }
    float lse_own = 0.f, del_own = 0.f;
    bool qv_own = true;
    if (MODE == 1) {
        lse_own = A.lse[bh * L + min(own_i, L - 1)];
        del_own = A.delta[bh * L + min(own_i, L - 1)];
        qv_own = qm ? (qm[min(own_i, L - 1)] != 0) : true;
    }
    const bool key_own_live = (MODE == 2) ? (own_in && (!km || km[min(own_i, S - 1)] != 0)) : true;
    for (int ob = w; ob * 32 < oth_len; ob += NW) {
        load_tile(Xoth, ldxt, ob * 32, oth_len, d, TS, oth_x, lane);
        load_tile(Yoth, ldyt, ob * 32, oth_len, d, TS, oth_y, lane);
        wave_lds_sync();
        bf32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
        {
            const float* ax = oth_x + l31 * TS + h;
            const float* bx = own_x + l31 * TS + h;
            const float* ay = oth_y + l31 * TS + h;
            const float* by = own_y + l31 * TS + h;
            for (int s2 = 0; s2 < d; s2 += 2) {
                sc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[s2], bx[s2], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[s2], by[s2], dp, 0, 0, 0);
            }
        }
        // P and dS in the score registers (register e <-> other row row_of(e, h), lane <-> own column)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oth_i = ob * 32 + row_of(r, h);
            float p = 0.f, ds = 0.f;
            if (MODE == 1) {                                            // own = query, other = key
                const bool dead = oth_i >= S || !own_in || (km && qv_own && !km[min(oth_i, S - 1)]);
                if (!dead) { p = expf(sc[r] * A.scale - lse_own); ds = A.scale * p * (dp[r] - del_own); }
            } else {                                                    // own = key, other = query
                const int qi = min(oth_i, L - 1);
                const bool qv = qm ? (qm[qi] != 0) : true;
                const bool dead = oth_i >= L || !own_in || (km && qv && !key_own_live);
                if (!dead) { p = expf(sc[r] * A.scale - A.lse[bh * L + qi]); ds = A.scale * p * (dp[r] - A.delta[bh * L + qi]); }
            }
            sc[r] = ds; dp[r] = p;
        }
        // G1^T[f][own] += X_oth^T dS;  (dK|dV) G2^T[f][own] += Y_oth^T P
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int off = row_of(e, h) * TS + 32 * i + l31;
                g1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(oth_x[off], sc[e], g1[i], 0, 0, 0);
                if (MODE == 2) g2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(oth_y[off], dp[e], g2[i], 0, 0, 0);
            }
        wave_lds_sync();
    }
    __syncthreads();                                                    // every wave is done with its tiles: the sums reuse them
    // ---- add the waves' accumulators in wave order, then store rows own0 .. own0 + 31, features 0 .. d - 1
    constexpr int GS = 33;                                              // [feature][own] with a padded stride
    float* buf1 = wave_base;
    float* buf2 = buf1 + DT * 32 * GS;
    for (int w0 = 0; w0 < NW; ++w0) {
        if (w == w0) {
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = (32 * i + row_of(r, h)) * GS + l31;
                    buf1[o] = (w0 == 0 ? 0.f : buf1[o]) + g1[i][r];
                    if (MODE == 2) buf2[o] = (w0 == 0 ? 0.f : buf2[o]) + g2[i][r];
                }
        }
        __syncthreads();
    }
    float* D1 = (MODE == 1 ? A.dq + (size_t)b * L * A.lddq : A.dk + (size_t)b * S * A.lddk) + hoff;
    const int ld1 = MODE == 1 ? A.lddq : A.lddk;
    float* D2 = MODE == 2 ? A.dv + (size_t)b * S * A.lddv + hoff : nullptr;
    for (int idx = t; idx < 32 * d; idx += 64 * NW) {
        const int o = idx / d, f = idx - o * d;
        if (own0 + o < own_len) {
            D1[(size_t)(own0 + o) * ld1 + f] = buf1[f * GS + o];
            if (MODE == 2) D2[(size_t)(own0 + o) * A.lddv + f] = buf2[f * GS + o];
        }
    }
}

template <int DT, int NW>
static size_t bwd_lds(int d) {
    const size_t tile = (size_t)32 * (d + 1) + 32;
    const size_t tiles = (2 + 2 * (size_t)NW) * tile;
    const size_t sums = 2 * (size_t)DT * 32 * 33;
    const size_t wave_area = 2 * (size_t)NW * tile;
    return (tiles + (sums > wave_area ? sums - wave_area : 0) + NW * 64 + 64) * sizeof(float);
}

template <int DT, int NW>
static int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
    const size_t lds = bwd_lds<DT, NW>(a.d);
    if (lds > 160 * 1024) return DR_ENOSUP;
    // raise the dynamic-LDS limit whenever this launch needs more than the largest size set so far for this instantiation (a later call with a
    // larger head dim inside the same DT bucket needs more than the first one did; as in sinkhorn.hip the attribute follows the need)
    static size_t lds_set = 0;
    if (lds > lds_set) {
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kernel<DT, NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kernel<DT, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DR_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kernel<DT, NW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    const dim3 blk(64 * NW), gq((a.L + 31) / 32, a.H, a.B), gk((a.S + 31) / 32, a.H, a.B);
    hipLaunchKernelGGL((attn_bwd_kernel<DT, NW, 0>), gq, blk, lds, st, a);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_kernel<DT, NW, 1>), gq, blk, lds, st, a);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL((attn_bwd_kernel<DT, NW, 2>), gk, blk, lds, st, a);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace
}  // namespace dr

using namespace dr;

extern "C" {

size_t dr_attention_backward_workspace_bytes(int B, int H, int L) { return (B > 0 && H > 0 && L > 0) ? (size_t)2 * B * H * L * sizeof(float) + 256 : 0; }

int dr_attention_backward_f32(int B, int H, int L, int S, int d, const float* q, const float* k, const float* v, const float* o, const float* grad_o,
                              int ld, const uint8_t* q_mask, const uint8_t* k_mask, float scale, float* grad_q, float* grad_k, float* grad_v,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || H < 1 || L < 1 || S < 1 || d < 4 || (d & 3) || ld < H * d || (ld & 3) || !q || !k || !v || !o || !grad_o || !grad_q || !grad_k || !grad_v)
        return DR_EINVAL;
    if ((q_mask == nullptr) != (k_mask == nullptr)) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!workspace || workspace_bytes < dr_attention_backward_workspace_bytes(B, H, L)) return DR_EWORKSPACE;
    AttnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.o = o; a.go = grad_o; a.dq = grad_q; a.dk = grad_k; a.dv = grad_v; a.qmask = q_mask; a.kmask = k_mask;
    a.lse = (float*)workspace; a.delta = a.lse + (size_t)B * H * L;
    a.ldq = a.ldk = a.ldv = a.ldo = a.ldg = a.lddq = a.lddk = a.lddv = ld;
    a.B = B; a.H = H; a.L = L; a.S = S; a.d = d; a.scale = scale;
    hipStream_t st = (hipStream_t)stream;
    if (d <= 64) return launch_bwd<2, 4>(a, st);
    if (d <= 96) return launch_bwd<3, 4>(a, st);
    if (d <= 128) return launch_bwd<4, 4>(a, st);
    if (d <= 160) return launch_bwd<5, 3>(a, st);
    return DR_ENOSUP;
}

/* forward of the same op on the inference kernels (attention.hip): out [B L, ld] = softmax(q k^T scale) v per head */
int dr_attention_f32(int B, int H, int L, int S, int d, const float* q, const float* k, const float* v, int ld, const uint8_t* q_mask,
                     const uint8_t* k_mask, float scale, float* out, void* stream) {
    if (B < 0 || H < 1 || L < 1 || S < 1 || d < 4 || (d & 3) || ld < H * d || (ld & 3) || !q || !k || !v || !out) return DR_EINVAL;
    if ((q_mask == nullptr) != (k_mask == nullptr)) return DR_EINVAL;
    if (B == 0) return DR_OK;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.out = out; a.ldq = a.ldk = a.ldv = a.ldo = ld; a.H = H; a.d = d;
    a.qmask = q_mask; a.kmask = k_mask;
    a.nseg = B; a.q0 = 0; a.qstride = L; a.Lq = L; a.k0 = 0; a.kstride = S; a.Lk = S; a.scale = scale;
    return launch_attention(a, (hipStream_t)stream);
}

}  // extern "C"
