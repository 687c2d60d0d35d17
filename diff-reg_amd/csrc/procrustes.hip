// procrustes.hip -- SoftProcrustesLayer on device: top-K of the N*M confidences, weighted Kabsch,
// 3x3 SVD in fp64, condition-number gate.  Replaces 3D/models/procrustes.py:17-93 including the
// per-step `.cpu().double().svd()` round trip (procrustes.py:35-36, quirk Q13): no host sync.
//
// One workgroup per pair:
//   1. the tile (<= 256 x 256) is read once into registers as ordered keys; radix select (4 x 8-bit
//      digits, run-length aggregated LDS atomics, parallel bin search) -> key of the K-th largest
//   2. entries > tau, plus the first entries == tau in a fixed (thread, element) order (torch.sort is
//      unstable, ties are unspecified upstream), feed the moment sums directly -- deterministic
//   3. fp64 sums  W1 = sum|w|, sum w X, sum w Y, sum w Y X^T  ->  Sxy = sum w^ Y X^T - (2 - s) Ybar Xbar^T
//      with w^ = w / (W1 + eps), s = sum w^  (identical to (Y - Ybar)^T (w^ (X - Xbar)), procrustes.py:27-33)
//   4. one-sided Jacobi SVD, R = U diag(1,1,det U det V) V^T, t = Ybar - R Xbar, cond = Dmax / Dmin
#include "kernels.h"
#include "svd3.h"

namespace dr {

constexpr int PK_MAX = 4096;   // largest K (= max(N, M) * sample_rate) supported

__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // larger float <=> larger key
}

struct ProcArgs {
    const float* conf;        // [P, N, M]
    const float* src_pcd;     // [P, N, 3]
    const float* tgt_pcd;     // [P, M, 3]
    const uint8_t* src_mask;  // [P, N] (4D variant: K from the mask sums) or nullptr
    const uint8_t* tgt_mask;
    float* R; float* t; float* Rf; float* tf;   // [P,9] [P,3] [P,9] [P,3]
    double* cond;             // [P]
    int* ok;                  // [P]
    int* topk_idx;            // optional [P, K] flat indices of the selected entries (a set: written in arrival order)
    int N, M, K_fixed, use_mask_len;
    float sample_rate, max_cond;
    // large tiles (N*M > 65536) with a workspace: the K selected entries, written in index order by proc_take_kernel
    // (nullptr = none: the fit kernel streams the tile itself)
    unsigned* g_ckey;         // [P, PC_CAP]
    int* g_cidx;              // [P, PC_CAP]
    int* g_ccount;            // [P] selected entries
    int S, SL;                // slices per tile, elements per slice (a multiple of 1024)
    // large tiles, exact chip-wide selection (proc_hist_kernel x 3 + proc_take_kernel): histograms of the keys' three digits
    unsigned* g_hist;         // [P, 3, PH_BINS]  digit histograms of the keys that match the digits above (atomics on integers)
    uint2* g_shist;           // [P, S, PH_LAST]  per slice and last digit b: {keys above (digits so far, b), keys equal to it}
};

constexpr int PH_BINS = 4096;    // digits of 12 / 12 / 8 bits (most significant first)
constexpr int PH_LAST = 256;
__host__ __device__ constexpr int ph_shift(int pass) { return pass == 0 ? 20 : (pass == 1 ? 8 : 0); }
__host__ __device__ constexpr unsigned ph_bins(int pass) { return pass == 2 ? (unsigned)PH_LAST : (unsigned)PH_BINS; }
constexpr int PC_CAP = 4096;     // capacity of a tile's list of selected entries (= PK_MAX)

__device__ __forceinline__ float key_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

constexpr int CAND_MAX = 4096;   // candidate list held in LDS

// radix select over the keys enumerated by `for_each` (4 x 8-bit digits, run-length aggregated LDS
// atomics, parallel bin search by wave 0): on return tau = key of the K-th largest, remaining = how
// many keys == tau belong to the top K.  Block-wide; K >= 1 and K <= number of enumerated keys.
template <int PASSES = 4, typename ForEach>
__device__ __forceinline__ void radix_select(ForEach&& for_each, unsigned K, unsigned* s_hist, unsigned* s_pr, unsigned& tau,
                                             unsigned& remaining) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned prefix = 0, mask = 0;
    remaining = K;
    for (int pass = 0; pass < PASSES; ++pass) {
        const int shift = 24 - 8 * pass;
        if (t < 256) s_hist[t] = 0;
        __syncthreads();
        int cur = -1;
        unsigned cnt = 0;
        for_each([&](unsigned k, int) {
            if (k != 0u && (k & mask) == prefix) {
                const int b = (int)((k >> shift) & 255u);
                if (b == cur) {
                    ++cnt;
                } else {
                    if (cnt) atomicAdd(&s_hist[cur], cnt);
                    cur = b;
                    cnt = 1;
                }
            }
        });
        if (cnt) atomicAdd(&s_hist[cur], cnt);
        __syncthreads();
        if (w == 0) {
            // lane l owns bins 255-4l .. 252-4l (descending); find the bin holding the `remaining`-th key
            unsigned h[4], c = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h[q] = s_hist[255 - 4 * lane - q]; c += h[q]; }
            unsigned incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const unsigned long long hit = __ballot(incl >= remaining);
            const int first = __ffsll((long long)hit) - 1;
            if (lane == first) {
                unsigned cum = incl - c;
                int q = 0;
                for (; q < 3; ++q) {
                    if (cum + h[q] >= remaining) break;
                    cum += h[q];
                }
                s_pr[0] = prefix | ((unsigned)(255 - 4 * lane - q) << shift);
                s_pr[1] = remaining - cum;
            }
        }
        __syncthreads();
        prefix = s_pr[0];
        remaining = s_pr[1];
        mask |= 0xFFu << shift;
    }
    tau = prefix;
}

// exclusive prefix (over the 1024 threads, thread order) of a per-thread count; also the block total
__device__ __forceinline__ int block_excl_scan(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = s_w[k];
        if (k < w) base += c;
        tot += c;
    }
    total = tot;
    return base + incl - v;
}

// ---- large tiles, exact selection with the whole chip (round 3) -------------------------------------------------------------
// (one workgroup streaming a 1024 x 2048 tile three times took 4.7 ms.)  Rounds 1-2 bounded the K-th largest entry from below with
// the K-th largest of 4096 cell maxima and compacted the candidates >= that bound for ONE workgroup to select from; that needs the
// candidates to fit a 4096-entry list, and flat or massively tied matrices (early denoising steps, the synthetic 2D-3D scenes)
// overflow it: the single workgroup then streamed the whole tile six times (460 us per call at 1024 x 2048).  Exact form: a
// 3-digit radix select (12 / 12 / 8 bits) whose histograms are built by S slice workgroups per tile (integer atomics:
// deterministic), every workgroup re-deriving the digits found so far from the completed histograms of the earlier launches.
// The last pass also leaves, PER
// SLICE and per value b of the last digit, how many of the slice's keys are above (digits so far, b) and how many equal it; the take
// pass reads those two numbers of every slice in front of its own at b = the K-th key's last digit, which gives it the position of
// its first selected entry in the tile's list and the number of ties already handed out (ties go in index order).  It writes exactly
// the K selected entries, in index order, into one contiguous list per tile: nothing is left for the fit kernel to select.
__device__ __forceinline__ int proc_topk_count(const ProcArgs& A, int pair, int* s_len) {
    const int t = threadIdx.x, N = A.N, M = A.M;
    int K = A.K_fixed;
    if (A.use_mask_len) {                                        // same rule as procrustes_kernel (quirk Q17)
        if (t < 2) s_len[t] = 0;
        __syncthreads();
        int c0 = 0, c1 = 0;
        for (int i = t; i < N; i += blockDim.x) c0 += A.src_mask[(size_t)pair * N + i] != 0;
        for (int j = t; j < M; j += blockDim.x) c1 += A.tgt_mask[(size_t)pair * M + j] != 0;
        if (c0) atomicAdd(&s_len[0], c0);
        if (c1) atomicAdd(&s_len[1], c1);
        __syncthreads();
        const int mx = s_len[0] > s_len[1] ? s_len[0] : s_len[1];
        K = (int)((float)mx * A.sample_rate);
    }
    const long NM = (long)N * M;
    if (K > NM) K = (int)NM;
    if (K > PK_MAX) K = PK_MAX;
    return K;
}
// the bin of a histogram (NB bins, larger bin = larger keys) that holds the `remaining`-th largest key, and the rank inside it;
// block-wide (1024 threads, h[] = the thread's NB / 1024 bins in descending order: bin NB-1-(t BPT + i)); s_out[0] = bin, s_out[1] = new
// remaining (bin 0, unchanged rank when the histogram holds fewer than `remaining` keys)
template <int NB, int BPT>
__device__ __forceinline__ void ph_find_bin(const unsigned (&h)[BPT], unsigned remaining, int* s_w, unsigned* s_out) {
    const int t = threadIdx.x;
    unsigned sum = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) sum += h[i];
    if (t == 0) { s_out[0] = 0u; s_out[1] = remaining; }
    int total;
    unsigned excl = (unsigned)block_excl_scan((int)sum, s_w, total);
    if (excl < remaining && excl + sum >= remaining) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            if (excl < remaining && excl + h[i] >= remaining) {
                s_out[0] = (unsigned)(NB - 1 - (t * BPT + i));
                s_out[1] = remaining - excl;
            }
            excl += h[i];
        }
    }
    __syncthreads();
}
// digits found by the passes before PASS: prefix (key bits above this pass's digit) and the rank still to find.  Every workgroup
// derives them from the completed histograms of the earlier launches; ALL their bins are fetched before the first scan so that the
// levels cost one load latency, not one each (a ticketed "last workgroup finds the digit" variant needs an agent-scope release and
// acquire per workgroup -- an L2 write-back and invalidate each -- and measured 1.5x slower per pass, 5x on 8 tiles).
template <int PASS>
__device__ __forceinline__ void ph_levels(const ProcArgs& A, int pair, unsigned K, int* s_w, unsigned* s_out, unsigned& prefix,
                                          unsigned& remaining) {
    constexpr int BPT = PH_BINS / 1024;
    const int t = threadIdx.x;
    const unsigned* gh = A.g_hist + (size_t)pair * 3 * PH_BINS;
    unsigned h0[BPT], h1[BPT], h2[1];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        h0[i] = PASS > 0 ? gh[PH_BINS - 1 - (t * BPT + i)] : 0u;
        h1[i] = PASS > 1 ? gh[PH_BINS + PH_BINS - 1 - (t * BPT + i)] : 0u;
    }
    h2[0] = (PASS > 2 && t < PH_LAST) ? gh[2 * PH_BINS + PH_LAST - 1 - t] : 0u;
    prefix = 0; remaining = K;
    if (PASS > 0) {
        ph_find_bin<PH_BINS, BPT>(h0, remaining, s_w, s_out);
        prefix |= s_out[0] << ph_shift(0); remaining = s_out[1];
        __syncthreads();
    }
    if (PASS > 1) {
        ph_find_bin<PH_BINS, BPT>(h1, remaining, s_w, s_out);
        prefix |= s_out[0] << ph_shift(1); remaining = s_out[1];
        __syncthreads();
    }
    if (PASS > 2) {
        ph_find_bin<PH_LAST, 1>(h2, remaining, s_w, s_out);
        prefix |= s_out[0] << ph_shift(2); remaining = s_out[1];
        __syncthreads();
    }
}
template <int PASS>
__global__ __launch_bounds__(1024) void proc_hist_kernel(ProcArgs A) {
    __shared__ unsigned s_h[PH_BINS];
    __shared__ int s_w[16];
    __shared__ unsigned s_out[2];
    __shared__ int s_len[2];
    __shared__ unsigned s_above;
    const int pair = blockIdx.y, sl = blockIdx.x, t = threadIdx.x;
    const int NM = A.N * A.M;
    const float* conf = A.conf + (size_t)pair * NM;
    // sweeps of 1024 entries, 8 at a time in registers; the first 8 loads are issued before the levels are derived (both wait on memory)
    const int s0 = sl * A.SL, s1 = min(NM, s0 + A.SL);
    auto load8 = [&](int c, unsigned (&k)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = s0 + (c * 8 + i) * 1024 + t;
            k[i] = e < s1 ? order_key(conf[e]) : 0u;
        }
    };
    unsigned ka[8], kb[8];
    load8(0, ka);
    const int K = proc_topk_count(A, pair, s_len);
    unsigned prefix, remaining;
    ph_levels<PASS>(A, pair, (unsigned)K, s_w, s_out, prefix, remaining);
    constexpr unsigned mask = PASS == 0 ? 0u : (PASS == 1 ? 0xFFF00000u : 0xFFFFFF00u);
    for (int b = t; b < PH_BINS; b += 1024) s_h[b] = 0u;
    if (t == 0) s_above = 0u;
    __syncthreads();
    constexpr int shift = ph_shift(PASS);
    constexpr unsigned bmask = ph_bins(PASS) - 1u;
    // run-length accumulation per thread: flat or heavily tied tiles put every key in one bin (1024 serialised LDS atomics per sweep)
    unsigned cur = 0xFFFFFFFFu, run = 0, above = 0;
    auto tally = [&](const unsigned (&kk)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned k = kk[i];
            if (PASS == 2) above += (k & mask) > prefix ? 1u : 0u;
            if (k != 0u && (k & mask) == prefix) {
                const unsigned b = (k >> shift) & bmask;
                if (b != cur) {
                    if (run) atomicAdd(&s_h[cur], run);
                    cur = b; run = 0;
                }
                ++run;
            }
        }
    };
    const int nchunk = (A.SL / 1024 + 7) / 8;
    for (int c = 0; c < nchunk; c += 2) {
        if (c + 1 < nchunk) load8(c + 1, kb);
        tally(ka);
        if (c + 1 < nchunk) {
            if (c + 2 < nchunk) load8(c + 2, ka);
            tally(kb);
        }
    }
    if (run) atomicAdd(&s_h[cur], run);
    if (PASS == 2) {
        above = wave_sum(above);
        if ((t & 63) == 0 && above) atomicAdd(&s_above, above);
    }
    __syncthreads();
    unsigned* gh = A.g_hist + ((size_t)pair * 3 + PASS) * PH_BINS;
    for (int b = t; b < (int)ph_bins(PASS); b += 1024) {
        const unsigned c = s_h[b];
        if (c) atomicAdd(&gh[b], c);
    }
    if (PASS == 2) {
        // per slice, per last digit b: {keys of the slice above (prefix, b), keys equal to it}; thread d owns bin PH_LAST-1-d
        const int bin = PH_LAST - 1 - t;
        const unsigned c = bin >= 0 ? s_h[bin] : 0u;
        int total;
        const unsigned gt = (unsigned)block_excl_scan((int)c, s_w, total) + s_above;
        if (bin >= 0) A.g_shist[((size_t)pair * A.S + sl) * PH_LAST + bin] = make_uint2(gt, c);
    }
}
// entries of a slice in (wave, sweep, lane) order: wave w owns the contiguous SL/16 entries from s0 + w SL/16 (SL is a multiple of 1024)
template <bool CACHED>
__global__ __launch_bounds__(1024) void proc_take_kernel(ProcArgs A) {
    __shared__ int s_w[16];
    __shared__ unsigned s_out[2];
    __shared__ int s_len[2];
    __shared__ unsigned s_wg[16], s_we[16];
    const int pair = blockIdx.y, sl = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int NM = A.N * A.M;
    const float* conf = A.conf + (size_t)pair * NM;
    const int WL = A.SL >> 4, IT = WL >> 6;                      // entries per wave, sweeps
    const int base = sl * A.SL + w * WL + lane;
    unsigned kc[CACHED ? 32 : 1];
    auto key_at = [&](int it) -> unsigned {
        const int e = base + it * 64;
        return e < NM ? order_key(conf[e]) : 0u;
    };
    if (CACHED) {                                                 // issued before the levels are derived (both wait on memory)
#pragma unroll
        for (int it = 0; it < 32; ++it) kc[it] = it < IT ? key_at(it) : 0u;
    }
    const int K = proc_topk_count(A, pair, s_len);
    unsigned tau, remaining;                                      // the K-th largest key; `remaining` of the entries equal to it are taken
    ph_levels<3>(A, pair, (unsigned)K, s_w, s_out, tau, remaining);
    // selected entries (above tau) and ties in the slices in front of this one
    unsigned gt_before = 0, eq_before = 0;
    for (int q = lane; q < sl; q += 64) {
        const uint2 v = A.g_shist[((size_t)pair * A.S + q) * PH_LAST + (tau & (PH_LAST - 1u))];
        gt_before += v.x; eq_before += v.y;
    }
    gt_before = wave_sum(gt_before);
    eq_before = wave_sum(eq_before);
    // f(key, sweep) over the wave's entries in order; beyond 32 sweeps the keys are re-read, 8 loads in flight
    auto sweep_all = [&](auto&& f) {
        if (CACHED) {
#pragma unroll
            for (int it = 0; it < 32; ++it) if (it < IT) f(kc[it], it);
        } else {
            for (int c = 0; c < IT; c += 8) {
                unsigned k8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) k8[i] = c + i < IT ? key_at(c + i) : 0u;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (c + i < IT) f(k8[i], c + i);
            }
        }
    };
    unsigned cg = 0, ce = 0;
    sweep_all([&](unsigned k, int) { cg += k > tau ? 1u : 0u; ce += (k == tau && k != 0u) ? 1u : 0u; });
    cg = wave_sum(cg);
    ce = wave_sum(ce);
    if (lane == 0) { s_wg[w] = cg; s_we[w] = ce; }
    __syncthreads();
    unsigned gtB = gt_before, eqB = eq_before, gtT = 0, eqT = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < w) { gtB += s_wg[k]; eqB += s_we[k]; }
        gtT += s_wg[k]; eqT += s_we[k];
    }
    const unsigned rem = K > 0 ? remaining : 0u;
    // position of this wave's first selected entry: everything above tau in front of it + the ties already handed out
    unsigned pos = gtB + (eqB < rem ? eqB : rem);
    unsigned* ck = A.g_ckey + (size_t)pair * PC_CAP;
    int* ci = A.g_cidx + (size_t)pair * PC_CAP;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto place = [&](unsigned k, int it) {
        const bool eq = k == tau && k != 0u;
        const unsigned long long beq = __ballot(eq);
        const bool sel = K > 0 && (k > tau || (eq && eqB + (unsigned)__popcll(beq & lt) < rem));
        const unsigned long long bsel = __ballot(sel);
        if (sel) {
            const unsigned o = pos + (unsigned)__popcll(bsel & lt);
            if (o < (unsigned)PC_CAP) { ck[o] = k; ci[o] = base + it * 64; }
        }
        pos += (unsigned)__popcll(bsel);
        eqB += (unsigned)__popcll(beq);
    };
    sweep_all(place);
    if (sl == A.S - 1 && t == 0) {
        const unsigned e_all = eq_before + eqT;
        A.g_ccount[pair] = K > 0 ? (int)(gt_before + gtT + (e_all < rem ? e_all : rem)) : 0;
    }
}

// REG: the tile (N*M <= 65536) is read ONCE and kept as 64 ordered keys per thread.  !REG: every pass
// streams the tile (L2).  Selection in two levels: the K-th largest of the 1024 per-thread maxima, L,
// is a lower bound of the K-th largest entry (there are >= K entries >= L), so only entries >= L --
// typically K..1.2 K of them -- are compacted into LDS and selected exactly; if they do not fit
// (flat or massively tied tiles) the full-tile radix select runs instead.
template <bool REG>
__global__ __launch_bounds__(1024) void procrustes_kernel(ProcArgs A) {
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_pr[2];
    __shared__ int s_w[16];
    __shared__ unsigned s_ckey[CAND_MAX];
    __shared__ int s_cidx[CAND_MAX];
    __shared__ double s_red[16][16];
    __shared__ int s_len[2];
    __shared__ int s_nsel;

    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int N = A.N, M = A.M, NM = N * M;
    const float* conf = A.conf + (size_t)pair * NM;

    unsigned key[REG ? 64 : 1];
    // register image: key[4 q + c] is element e = 4 (1024 q + t) + c  (16 coalesced 16-byte loads per thread)
    const bool vec4 = REG && (NM % 4 == 0) && (((uintptr_t)conf & 15) == 0);
    if (REG) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = 4 * (q * 1024 + t);
            if (vec4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < NM) v = *reinterpret_cast<const float4*>(conf + e);
                key[4 * q + 0] = e < NM ? order_key(v.x) : 0u;      // 0 = "no element" (only one NaN pattern maps to key 0)
                key[4 * q + 1] = e < NM ? order_key(v.y) : 0u;
                key[4 * q + 2] = e < NM ? order_key(v.z) : 0u;
                key[4 * q + 3] = e < NM ? order_key(v.w) : 0u;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) key[4 * q + c] = e + c < NM ? order_key(conf[e + c]) : 0u;
            }
        }
    }
    auto for_each = [&](auto&& f) {
        if (REG) {
#pragma unroll
            for (int i = 0; i < 64; ++i) f(key[i], 4 * ((i >> 2) * 1024 + t) + (i & 3));
        } else {
            for (int e = t; e < NM; e += 1024) f(order_key(conf[e]), e);
        }
    };

    // ---- K (procrustes.py:61-65; 4D/models/procrustes.py:61-62 uses the mask sums, quirk Q17) -------
    int K = A.K_fixed;
    if (t == 0) s_nsel = 0;
    if (A.use_mask_len) {
        if (t < 2) s_len[t] = 0;
        __syncthreads();
        int c0 = 0, c1 = 0;
        for (int i = t; i < N; i += 1024) c0 += A.src_mask[(size_t)pair * N + i] != 0;
        for (int j = t; j < M; j += 1024) c1 += A.tgt_mask[(size_t)pair * M + j] != 0;
        if (c0) atomicAdd(&s_len[0], c0);
        if (c1) atomicAdd(&s_len[1], c1);
        __syncthreads();
        const int mx = s_len[0] > s_len[1] ? s_len[0] : s_len[1];
        K = (int)((float)mx * A.sample_rate);
    }
    if (K > NM) K = NM;
    if (K > PK_MAX) K = PK_MAX;
    __syncthreads();

    int n_listed = 0;
    // the sums are zeroed only when the selection is done: 32 live registers less while the 64 keys are
    double acc[16];
    const float* Xs = A.src_pcd + (size_t)pair * N * 3;
    const float* Ys = A.tgt_pcd + (size_t)pair * M * 3;
    auto take = [&](unsigned k, int e) {
        const double wv = (double)key_value(k);
        const int i = e / M, j = e - i * M;
        const double x0 = Xs[i * 3], x1 = Xs[i * 3 + 1], x2 = Xs[i * 3 + 2];
        const double y0 = Ys[j * 3], y1 = Ys[j * 3 + 1], y2 = Ys[j * 3 + 2];
        acc[0] += fabs(wv);
        acc[1] += wv * x0; acc[2] += wv * x1; acc[3] += wv * x2;
        acc[4] += wv * y0; acc[5] += wv * y1; acc[6] += wv * y2;
        acc[7] += wv * y0 * x0; acc[8] += wv * y0 * x1; acc[9] += wv * y0 * x2;
        acc[10] += wv * y1 * x0; acc[11] += wv * y1 * x1; acc[12] += wv * y1 * x2;
        acc[13] += wv * y2 * x0; acc[14] += wv * y2 * x1; acc[15] += wv * y2 * x2;
        if (A.topk_idx) A.topk_idx[(size_t)pair * A.K_fixed + atomicAdd(&s_nsel, 1)] = e;     // (row stride = the caller's K, also where a pair's own K is smaller: 4D)
    };
    if (K > 0) {
        // ---- level 1: lower bound L from the per-thread maxima ---------------------------------------------
        unsigned L = 0, dummy;
        int ncand = NM;
        if (!REG && A.g_ckey) {
            // the K selected entries were written by proc_take_kernel, in index order
            const int tot = A.g_ccount[pair];
            if (tot == K && tot <= CAND_MAX) {
                ncand = tot;
                for (int c = t; c < tot; c += 1024) {
                    s_ckey[c] = A.g_ckey[(size_t)pair * PC_CAP + c];
                    s_cidx[c] = A.g_cidx[(size_t)pair * PC_CAP + c];
                }
            }
            __syncthreads();
        } else if (K <= 1024) {
            // (threads whose slice is empty have tmax = 0 and are not enumerated; K <= #non-empty is
            //  guaranteed when K <= min(NM, 1024) because slices are filled round-robin)
            // L = the K-th largest thread maximum, all four digits (round 3: the 16-bit bucket of rounds 1-2 admitted the whole
            // near-uniform background of the sharp late-step matrices -- K-th falls into it as soon as two large entries share a
            // thread -- and sent the last four steps of every run to the 60 us dearer exact select below; +1.1 us here)
            unsigned tmax = 0;
            for_each([&](unsigned k, int) { tmax = k > tmax ? k : tmax; });
            radix_select<4>([&](auto&& f) { f(tmax, 0); }, (unsigned)K, s_hist, s_pr, L, dummy);
            int c = 0;
            for_each([&](unsigned k, int) { c += (k >= L && k != 0u) ? 1 : 0; });
            int off = block_excl_scan(c, s_w, ncand);
            if (ncand <= CAND_MAX) {
                for_each([&](unsigned k, int e) {
                    if (k >= L && k != 0u) { s_ckey[off] = k; s_cidx[off] = e; ++off; }
                });
            }
            __syncthreads();
        } else {
            // K > 1024 (thin tiles such as 2048 x 2, or a large sample_rate): the per-thread maxima do not bound the K-th largest
            // entry, so no candidate list is built -- take the full-tile radix path whatever the tile's size (without this the
            // list would be read uninitialised when N M <= CAND_MAX)
            ncand = CAND_MAX + 1;
        }
        unsigned tau = 0, remaining = 0;
        if (ncand > CAND_MAX) {
            // ---- fallback: exact select over the whole tile, then compact the K selected entries ------------
            auto for_mem = [&](auto&& f) {
                for (int e = t; e < NM; e += 1024) f(order_key(conf[e]), e);
            };
            unsigned ftau = 0, frem;
            int cg = 0, ce = 0;
            if (REG) {
                // the K-th largest key by bisection on its bits, from the registers: ftau = the largest v with #(key >= v) >= K.
                // 32 rounds of 64 wave-wide compares, one barrier each, whatever the values
                // (the digit histograms of radix_select serialise on a few hot LDS bins for exactly the tiles that end up here:
                // sharp late-step matrices whose K large entries share threads, flat early ones; it cost 75 us)
                unsigned* s_cnt = s_hist;                         // [2][16] wave counts, double-buffered
                for (int b = 31; b >= 0; --b) {
                    const unsigned cand = ftau | (1u << b);
                    // half of the keys counted by the scalar unit (lane mask -> s_bcnt1 -> s_add: the CU's ONE scalar unit serves all
                    // 16 waves), half by the vector unit (compare + add-with-carry per lane, one wave sum per round)
                    unsigned c = 0, cv = 0;
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        c += (unsigned)__popcll(__ballot(key[i] >= cand));
                        cv += key[32 + i] >= cand ? 1u : 0u;
                        // (left alone the counts sink into the `lane == 0` branch below and the lane masks they need are kept
                        //  alive through v_writelane spills: pin the running count to a scalar register every 8 keys)
                        if ((i & 7) == 7) asm volatile("" : "+s"(c));
                    }
                    c += wave_sum(cv);
                    if (lane == 0) s_cnt[(b & 1) * 16 + w] = c;
                    __syncthreads();
                    unsigned tot = 0;
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) tot += s_cnt[(b & 1) * 16 + k2];
                    if (tot >= (unsigned)K) ftau = cand;
                }
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    cg += key[i] > ftau ? 1 : 0;
                    ce += (key[i] == ftau && key[i] != 0u) ? 1 : 0;
                }
            } else {
                radix_select(for_mem, (unsigned)K, s_hist, s_pr, ftau, frem);
                for_mem([&](unsigned k, int) {
                    cg += k > ftau ? 1 : 0;
                    ce += (k == ftau && k != 0u) ? 1 : 0;
                });
            }
            int tg, te;
            int off_g = block_excl_scan(cg, s_w, tg);
            int rank_e = block_excl_scan(ce, s_w, te);
            frem = (unsigned)(K - tg);
            // (the same element-to-thread assignment as the counts above: the registers' when REG)
            auto for_own = [&](auto&& f) {
                if (REG) for_each(f);
                else for_mem(f);
            };
            for_own([&](unsigned k, int e) {
                if (k >= ftau && k != 0u) {                       // rare (K of the tile's entries) unless the tile is flat
                    const bool g = k > ftau;
                    const int pos = g ? off_g : tg + rank_e;
                    if (g || rank_e < (int)frem) { s_ckey[pos] = k; s_cidx[pos] = e; }
                    off_g += g ? 1 : 0;                           // (selects: incrementing one of two counters per branch sent both to scratch)
                    rank_e += g ? 0 : 1;
                }
            });
            ncand = K;                      // the list now holds exactly the top K: select all of it below
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0;
        // ---- exact selection among the listed candidates (thread t owns entries t, t+1024, ..) ----------------
        auto for_cand = [&](auto&& f) {
            for (int c = t; c < ncand; c += 1024) f(s_ckey[c], s_cidx[c]);
        };
        n_listed = ncand;
        if (ncand > K) radix_select(for_cand, (unsigned)K, s_hist, s_pr, tau, remaining);
        int ties = 0;
        for_cand([&](unsigned k, int e) {
            if (k > tau) take(k, e);
            else if (k == tau) ++ties;
        });
        int tot_ties;
        int rank = block_excl_scan(ties, s_w, tot_ties);
        for_cand([&](unsigned k, int e) {
            if (k == tau) {
                if (rank < (int)remaining) take(k, e);
                ++rank;
            }
        });
    }
    if (K <= 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    }
    // only the waves that own list entries (entry c belongs to thread c % 1024) hold non-zero sums
    if (w * 64 < n_listed) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double v = wave_sum(acc[i]);
            if (lane == 0) s_red[w][i] = v;
        }
    } else if (lane < 16) {
        s_red[w][lane] = 0.0;
    }
    __syncthreads();
    // sum i over the waves by thread i (one thread reading all 256 partial sums at once spilled them to scratch: 6-11 us)
    if (t < 16) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += s_red[k][t];
        s_red[0][t] = v;
    }
    __syncthreads();
    if (t != 0) return;
    double sum[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] = s_red[0][i];
    const double inv = 1.0 / (sum[0] + 1e-4);        // eps of batch_weighted_procrustes
    const double sw = sum[0] * inv;                   // sum of normalised weights (w >= 0 here)
    double mx[3] = {sum[1] * inv, sum[2] * inv, sum[3] * inv};
    double my[3] = {sum[4] * inv, sum[5] * inv, sum[6] * inv};
    double Sxy[3][3], U[3][3], V[3][3], D[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Sxy[a][b] = sum[7 + 3 * a + b] * inv - (2.0 - sw) * my[a] * mx[b];
    // the reference forms Sxy in fp32 before `.double()` (procrustes.py:33-34)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Sxy[a][b] = (double)(float)Sxy[a][b];
    double Aw[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Aw[a][b] = Sxy[a][b];
    svd3_jacobi(Aw, U, D, V);
    const double cond = D[0] / D[2];
    const double dd = det3(U) * det3(V);
    double Rm[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Rm[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + dd * U[a][2] * V[b][2];
    float Rf32[9], tf32[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Rf32[a * 3 + b] = (float)Rm[a][b];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // t = mean_Y - R mean_X in fp32 like the reference (procrustes.py:43)
        const float mxf[3] = {(float)mx[0], (float)mx[1], (float)mx[2]};
        float dot = Rf32[a * 3] * mxf[0];
        dot = fmaf(Rf32[a * 3 + 1], mxf[1], dot);
        dot = fmaf(Rf32[a * 3 + 2], mxf[2], dot);
        tf32[a] = (float)my[a] - dot;
    }
    const bool good = cond < (double)A.max_cond;     // NaN or inf -> false (procrustes.py:87)
    float* R = A.R + (size_t)pair * 9; float* tt = A.t + (size_t)pair * 3;
    float* Rf = A.Rf + (size_t)pair * 9; float* tf = A.tf + (size_t)pair * 3;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        R[i] = Rf32[i];
        Rf[i] = good ? Rf32[i] : ((i % 4 == 0) ? 1.f : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        tt[i] = tf32[i];
        tf[i] = good ? tf32[i] : 0.f;
    }
    A.cond[pair] = cond;
    A.ok[pair] = good ? 1 : 0;
}

// slices of a large tile: a workgroup of the selection passes each, about 512 of them over the batch (all co-resident: the levels'
// preamble is paid once, not once per round), at most 256 per tile, of a multiple of 1024 entries (32768 or fewer keep the take pass's
// keys in registers)
static int proc_slice_len(int P, long NM) {
    long S = 512 / (P > 0 ? P : 1);
    S = S < 1 ? 1 : (S > 256 ? 256 : S);
    long SL = (NM + S - 1) / S;
    SL = (SL + 1023) / 1024 * 1024;
    return (int)(SL < 4096 ? 4096 : SL);
}
static size_t proc_zeroed_bytes(int P) { return (size_t)P * 3 * PH_BINS * 4; }
size_t procrustes_workspace_bytes(int P, int N, int M) {
    const long NM = (long)N * M;
    if (P <= 0 || NM <= 65536) return 0;
    const long SL = proc_slice_len(P, NM);
    const size_t S = (size_t)((NM + SL - 1) / SL);
    // the list (keys, indices), its length, digit histograms, per-slice {above, equal} tables
    return (size_t)P * (PC_CAP * 8 + S * PH_LAST * 8) + (((size_t)P * 4 + 255) & ~(size_t)255) + proc_zeroed_bytes(P) + 512;
}

int launch_procrustes(const float* conf, const float* src_pcd, const float* tgt_pcd, const uint8_t* src_mask,
                      const uint8_t* tgt_mask, int P, int N, int M, int use_mask_len, float sample_rate, float max_cond,
                      float* R, float* t, float* Rf, float* tf, double* cond, int* ok, int* topk_idx, hipStream_t st,
                      void* ws, size_t ws_bytes) {
    if (P <= 0) return DR_OK;
    if ((long)N * M > 0x7fffffffL) return DR_ENOSUP;
    ProcArgs a;
    a.conf = conf; a.src_pcd = src_pcd; a.tgt_pcd = tgt_pcd; a.src_mask = src_mask; a.tgt_mask = tgt_mask;
    a.R = R; a.t = t; a.Rf = Rf; a.tf = tf; a.cond = cond; a.ok = ok; a.topk_idx = topk_idx;
    a.N = N; a.M = M; a.sample_rate = sample_rate; a.max_cond = max_cond;
    a.use_mask_len = (use_mask_len && src_mask && tgt_mask) ? 1 : 0;
    // K = int(int(max(len_s, len_t) * rate))  with float32 arithmetic (procrustes.py:63-65)
    a.K_fixed = (int)((float)(N > M ? N : M) * sample_rate);
    if (a.K_fixed > PK_MAX) return DR_ENOSUP;
    a.g_ckey = nullptr; a.g_cidx = nullptr; a.g_ccount = nullptr; a.g_hist = nullptr; a.g_shist = nullptr; a.S = a.SL = 0;
    ProfScope ps(PK_PROCRUSTES, (double)P * N * M * 4.0, st);
    const long NM = (long)N * M;
    if (NM <= 65536) {
        hipLaunchKernelGGL(procrustes_kernel<true>, dim3(P), dim3(1024), 0, st, a);
    } else {
        if (ws && ws_bytes >= procrustes_workspace_bytes(P, N, M)) {
            a.SL = proc_slice_len(P, NM);
            a.S = (int)((NM + a.SL - 1) / a.SL);
            char* w8 = (char*)ws;
            a.g_ckey = (unsigned*)w8; w8 += (size_t)P * PC_CAP * 4;
            a.g_cidx = (int*)w8; w8 += (size_t)P * PC_CAP * 4;
            a.g_ccount = (int*)w8; w8 += ((size_t)P * 4 + 255) & ~(size_t)255;
            a.g_hist = (unsigned*)w8;
            DR_HIP_CHECK(hipMemsetAsync(a.g_hist, 0, proc_zeroed_bytes(P), st));
            w8 += proc_zeroed_bytes(P);
            a.g_shist = (uint2*)w8;
            hipLaunchKernelGGL(proc_hist_kernel<0>, dim3(a.S, P), dim3(1024), 0, st, a);
            DR_LAUNCH_CHECK();
            hipLaunchKernelGGL(proc_hist_kernel<1>, dim3(a.S, P), dim3(1024), 0, st, a);
            DR_LAUNCH_CHECK();
            hipLaunchKernelGGL(proc_hist_kernel<2>, dim3(a.S, P), dim3(1024), 0, st, a);
            DR_LAUNCH_CHECK();
            if (a.SL <= 32768) hipLaunchKernelGGL(proc_take_kernel<true>, dim3(a.S, P), dim3(1024), 0, st, a);
            else hipLaunchKernelGGL(proc_take_kernel<false>, dim3(a.S, P), dim3(1024), 0, st, a);
            DR_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(procrustes_kernel<false>, dim3(P), dim3(1024), 0, st, a);
    }
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace dr
