// stateops.hip -- element-wise / reduction kernels on the N x M diffusion state (HBM-bound).
//   pair_min      x.min() per pair                          (3D/models/pipeline.py:239,264)
//   ddim_update   predict_noise_from_start + DDIM step      (pipeline.py:246-256, 287-291; 4D/...:188-190)
//   sigmoid       4D read-out                               (4D/models/pipeline.py:192)
//   top1_union    mutual_topk_select(k=1, mutual=False)     (pipeline.py:12-65, 275-280)
#include "kernels.h"

namespace dr {

// ---------------------------------------------------------------------------------------------
// sm / tm (nullable, [P,N] / [P,M]): the minimum runs over the entries inside both masks only (DR_LOOP_RAGGED)
__global__ __launch_bounds__(1024) void pair_min_kernel(const double* __restrict__ x, int NM, int M, const uint8_t* __restrict__ sm,
                                                        const uint8_t* __restrict__ tm, double* __restrict__ out) {
    __shared__ double s[16];
    const double* p = x + (size_t)blockIdx.x * NM;
    double m = INFINITY;
    if (sm) {
        const uint8_t* s1 = sm + (size_t)blockIdx.x * (NM / M);
        const uint8_t* t1 = tm + (size_t)blockIdx.x * M;
        for (int e = threadIdx.x; e < NM; e += 1024)
            if (s1[e / M] && t1[e % M]) m = fmin(m, p[e]);
    } else
    for (int e = threadIdx.x; e < NM; e += 1024) m = fmin(m, p[e]);
    m = wave_min(m);
    if (lane_id() == 0) s[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = s[0];
        for (int k = 1; k < 16; ++k) r = fmin(r, s[k]);
        out[blockIdx.x] = r;
    }
}

// The same minimum with G workgroups per pair (a single pair, or a few large ones: one workgroup per pair is then one CU streaming
// the whole matrix -- 88 us at 564 x 629): slice minima to `part`, and the workgroup that arrives last at the pair's counter reduces
// them (minimum is exact: the result does not depend on the order).  `cnt` must be zero on entry and is left zero.
constexpr int PM_G = 64;
__global__ __launch_bounds__(256) void pair_min_split_kernel(const double* __restrict__ x, int NM, int M, const uint8_t* __restrict__ sm,
                                                             const uint8_t* __restrict__ tm, double* __restrict__ part,
                                                             unsigned* __restrict__ cnt, double* __restrict__ out) {
    __shared__ double s[4];
    __shared__ bool last;
    const int pair = blockIdx.y, G = gridDim.x;
    const double* p = x + (size_t)pair * NM;
    const int per = ((NM + G - 1) / G + 255) / 256 * 256, e0 = blockIdx.x * per, e1 = min(NM, e0 + per);
    double m = INFINITY;
    if (sm) {
        const uint8_t* s1 = sm + (size_t)pair * (NM / M);
        const uint8_t* t1 = tm + (size_t)pair * M;
        for (int e = e0 + threadIdx.x; e < e1; e += 256)
            if (s1[e / M] && t1[e % M]) m = fmin(m, p[e]);
    } else
        for (int e = e0 + threadIdx.x; e < e1; e += 256) m = fmin(m, p[e]);
    m = wave_min(m);
    if (lane_id() == 0) s[wave_id()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[(size_t)pair * PM_G + blockIdx.x] = fmin(fmin(s[0], s[1]), fmin(s[2], s[3]));
        __threadfence();                                          // the slice minimum is visible device-wide before the arrival is
        last = atomicAdd(&cnt[pair], 1u) == (unsigned)(G - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 64) {
        double r = INFINITY;
        for (int g = threadIdx.x; g < G; g += 64)
            r = fmin(r, __hip_atomic_load(&part[(size_t)pair * PM_G + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        r = wave_min(r);
        if (threadIdx.x == 0) { out[pair] = r; cnt[pair] = 0u; }
    }
}

size_t pair_min_scratch_bytes(int P) { return (size_t)(P > 0 ? P : 1) * (PM_G * sizeof(double) + 64); }

int launch_pair_min(const double* x, int P, int NM, double* out, hipStream_t st, int M, const uint8_t* sm, const uint8_t* tm, void* scratch) {
    if (P <= 0) return DR_OK;
    ProfScope ps(PK_STATE, (double)P * NM * 8.0, st);
    if (!(sm && tm && M > 0)) { sm = tm = nullptr; M = 1; }
    if (scratch && P <= 32 && NM >= 16384) {
        const int G = min(PM_G, (NM + 4095) / 4096);
        double* part = (double*)scratch;
        unsigned* cnt = (unsigned*)((char*)scratch + (size_t)P * PM_G * sizeof(double));
        hipLaunchKernelGGL(pair_min_split_kernel, dim3(G, P), dim3(256), 0, st, x, NM, M, sm, tm, part, cnt, out);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    hipLaunchKernelGGL(pair_min_kernel, dim3(P), dim3(1024), 0, st, x, NM, M, sm, tm, out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---------------------------------------------------------------------------------------------
// x <- x0 * sqrt(a_next) + c * ((sra * xs - x0) / srm1) [+ sigma * xi],  xs = x - shift (3D) or x (4D)
// dtype bookkeeping of the reference (quirk Q2): the state is float32 on the first step, so
// xs = float32(x) - float32(min) and sigma*xi are float32 operations there; x0*sqrt(a_next) is a
// float32 product on every step (0-d float64 tensors do not promote); everything else is float64.
// Masked entries (x was filled with -inf in place, pipeline.py:296) stay -inf.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void ddim_kernel(DdimArgs A) {
    const int NM = A.N * A.M;
    const int pair = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= NM) return;
    const size_t g = (size_t)pair * NM + e;
    double xs = A.x[g];
    if (A.shift) {
        if (A.first_step) xs = (double)((float)xs - (float)A.shift[pair]);
        else xs = xs - A.shift[pair];
    }
    bool masked = false;
    if (A.src_mask) masked = !A.src_mask[(size_t)pair * A.N + e / A.M] || !A.tgt_mask[(size_t)pair * A.M + e % A.M];
    const float x0 = A.x0[g];
    const double eps = (A.sra * xs - (double)x0) / A.srm1;
    double xn = (double)(x0 * A.sqrt_an) + A.c * eps;
    if (A.noise) {
        if (A.first_step) xn = xn + (double)((float)A.sigma * A.noise[g]);
        else xn = xn + A.sigma * (double)A.noise[g];
    }
    A.x[g] = masked ? -INFINITY : xn;
}

int launch_ddim(const DdimArgs& a, int P, hipStream_t st) {
    if (P <= 0) return DR_OK;
    const int NM = a.N * a.M;
    ProfScope ps(PK_STATE, (double)P * NM * 20.0, st);
    hipLaunchKernelGGL(ddim_kernel, dim3((NM + 255) / 256, P), dim3(256), 0, st, a);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_to_f64_kernel(const float* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
int launch_f32_to_f64(const float* in, double* out, size_t n, hipStream_t st) {
    if (!n) return DR_OK;
    hipLaunchKernelGGL(f32_to_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

__global__ __launch_bounds__(256) void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
int launch_f64_to_f32(const double* in, float* out, size_t n, hipStream_t st) {
    if (!n) return DR_OK;
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.0 / (1.0 + exp(-in[i]));
}
int launch_sigmoid(const double* in, double* out, size_t n, hipStream_t st) {
    if (!n) return DR_OK;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---------------------------------------------------------------------------------------------
// top-1 union: True at (i, argmax_j conf_ij) for every row and (argmax_i conf_ij, j) for every
// column, listed row-major as [0, i, j] (int64).  The <= N + M hits are bitonic-sorted by flat index
// in LDS and de-duplicated (one workgroup per pair).  First occurrence wins a tie.
// With a workspace the arg-maxima come from the whole chip: workgroups of T1_ROWS rows leave the row
// arg-maxima and, per column, the arg-maximum over their rows; the pair's workgroup merges the row
// blocks in order.  (One workgroup reading a 1024 x 2048 f64 tile twice took 2.0 ms, 12% of the
// 2D-3D loop; a 256 x 256 tile 144 us.)
// ---------------------------------------------------------------------------------------------
constexpr int T1_MAX = 4096;   // N + M <= T1_MAX
constexpr int T1_ROWS = 16;    // rows per workgroup of the chip-wide pass (one wave per row)

// s_key[0 .. N+M) = flat indices of the hits (0xFFFFFFFF = none): sort, drop duplicates, write [0, i, j] rows and the count
__device__ __forceinline__ void top1_emit(unsigned* s_key, int* s_w, int N, int M, long long* __restrict__ o, int* __restrict__ cnt) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int n2 = 1;
    while (n2 < N + M) n2 <<= 1;
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < n2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = s_key[i], b = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // ordered compaction of the distinct keys: thread t owns entries 4t .. 4t+3 (T1_MAX = 4 x 1024)
    unsigned k4[4];
    bool keep[4];
    int c = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * t + q;
        k4[q] = i < n2 ? s_key[i] : 0xFFFFFFFFu;
        const unsigned prev = (i > 0 && i < n2) ? s_key[i - 1] : 0xFFFFFFFFu;
        keep[q] = k4[q] != 0xFFFFFFFFu && (i == 0 || k4[q] != prev);
        c += keep[q] ? 1 : 0;
    }
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int v = s_w[k];
        if (k < w) base += v;
        tot += v;
    }
    int n = base + incl - c;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (keep[q]) {
            o[n * 3] = 0; o[n * 3 + 1] = k4[q] / M; o[n * 3 + 2] = k4[q] % M;
            ++n;
        }
    if (t == 0) *cnt = tot;
}

template <typename T>
__global__ __launch_bounds__(1024) void top1_union_kernel(const T* __restrict__ conf, int N, int M, long long* __restrict__ out,
                                                          int* __restrict__ count, const uint8_t* __restrict__ smask,
                                                          const uint8_t* __restrict__ tmask) {
    __shared__ unsigned s_key[T1_MAX];
    __shared__ int s_w[16];
    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const T* c = conf + (size_t)pair * N * M;
    // masks (nullable): rows / columns outside them do not exist (DR_LOOP_RAGGED)
    const uint8_t* sm = smask ? smask + (size_t)pair * N : nullptr;
    const uint8_t* tm = tmask ? tmask + (size_t)pair * M : nullptr;
    for (int i = t; i < T1_MAX; i += 1024) s_key[i] = 0xFFFFFFFFu;
    __syncthreads();
    // row arg-max: one wave per row
    for (int i = w; i < N; i += 16) {
        if (sm && !sm[i]) continue;
        T best = -INFINITY; int bj = 0x7fffffff;
        for (int j = lane; j < M; j += 64) {
            if (tm && !tm[j]) continue;
            const T v = c[(size_t)i * M + j];
            if (v > best || (v == best && j < bj) || bj == 0x7fffffff) { best = v; bj = j; }
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const T ov = __shfl_xor(best, m); const int oj = __shfl_xor(bj, m);
            if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        if (lane == 0) s_key[i] = (unsigned)(i * M + bj);
    }
    // column arg-max: one thread per column (coalesced over j)
    for (int j = t; j < M; j += 1024) {
        if (tm && !tm[j]) continue;
        T best = -INFINITY; int bi = -1;
        for (int i = 0; i < N; ++i) {
            if (sm && !sm[i]) continue;
            const T v = c[(size_t)i * M + j];
            if (v > best || bi < 0) { best = v; bi = i; }
        }
        if (bi >= 0) s_key[N + j] = (unsigned)(bi * M + j);
    }
    __syncthreads();
    top1_emit(s_key, s_w, N, M, out + (size_t)pair * (N + M) * 3, count + pair);
}

// chip-wide pass: workgroup (rb, pair) owns rows [rb T1_ROWS, ..+T1_ROWS): same comparisons as above, restricted to its rows
template <typename T>
__global__ __launch_bounds__(1024) void top1_block_kernel(const T* __restrict__ conf, int N, int M, unsigned* __restrict__ rowkey,
                                                          T* __restrict__ colval, int* __restrict__ colrow,
                                                          const uint8_t* __restrict__ smask, const uint8_t* __restrict__ tmask) {
    const int pair = blockIdx.y, rb = blockIdx.x, NB = gridDim.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const T* c = conf + (size_t)pair * N * M;
    const uint8_t* sm = smask ? smask + (size_t)pair * N : nullptr;
    const uint8_t* tm = tmask ? tmask + (size_t)pair * M : nullptr;
    const int i0 = rb * T1_ROWS, i1 = min(N, i0 + T1_ROWS);
    {
        const int i = i0 + w;
        if (i < i1) {
            unsigned key = 0xFFFFFFFFu;
            if (!(sm && !sm[i])) {
                T best = -INFINITY; int bj = 0x7fffffff;
                for (int j = lane; j < M; j += 64) {
                    if (tm && !tm[j]) continue;
                    const T v = c[(size_t)i * M + j];
                    if (v > best || (v == best && j < bj) || bj == 0x7fffffff) { best = v; bj = j; }
                }
                for (int m = 32; m >= 1; m >>= 1) {
                    const T ov = __shfl_xor(best, m); const int oj = __shfl_xor(bj, m);
                    if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
                }
                key = (unsigned)(i * M + bj);
            }
            if (lane == 0) rowkey[(size_t)pair * N + i] = key;
        }
    }
    for (int j = t; j < M; j += 1024) {
        T best = -INFINITY; int bi = -1;
        if (!(tm && !tm[j])) {
            T v[T1_ROWS];
#pragma unroll
            for (int r = 0; r < T1_ROWS; ++r) v[r] = i0 + r < i1 ? c[(size_t)(i0 + r) * M + j] : (T)0;
#pragma unroll
            for (int r = 0; r < T1_ROWS; ++r) {
                if (i0 + r >= i1 || (sm && !sm[i0 + r])) continue;
                if (v[r] > best || bi < 0) { best = v[r]; bi = i0 + r; }
            }
        }
        colval[((size_t)pair * NB + rb) * M + j] = best;
        colrow[((size_t)pair * NB + rb) * M + j] = bi;
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void top1_merge_kernel(int N, int M, int NB, const unsigned* __restrict__ rowkey,
                                                          const T* __restrict__ colval, const int* __restrict__ colrow,
                                                          long long* __restrict__ out, int* __restrict__ count) {
    __shared__ unsigned s_key[T1_MAX];
    __shared__ int s_w[16];
    const int pair = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < T1_MAX; i += 1024) s_key[i] = i < N ? rowkey[(size_t)pair * N + i] : 0xFFFFFFFFu;
    __syncthreads();
    for (int j = t; j < M; j += 1024) {
        T best = -INFINITY; int bi = -1;
        for (int q = 0; q < NB; ++q) {                           // row blocks in order: the first row that attains the maximum wins
            const T v = colval[((size_t)pair * NB + q) * M + j];
            const int r = colrow[((size_t)pair * NB + q) * M + j];
            if (r >= 0 && (v > best || bi < 0)) { best = v; bi = r; }
        }
        if (bi >= 0) s_key[N + j] = (unsigned)(bi * M + j);
    }
    __syncthreads();
    top1_emit(s_key, s_w, N, M, out + (size_t)pair * (N + M) * 3, count + pair);
}

size_t top1_union_workspace_bytes(int P, int N, int M, size_t elt) {
    if (P <= 0 || N < 2 * T1_ROWS) return 0;                     // a single row block: the one-workgroup kernel is the same work
    const size_t NB = (size_t)(N + T1_ROWS - 1) / T1_ROWS;
    return (size_t)P * NB * M * (elt + 4) + (((size_t)P * N * 4 + 15) & ~(size_t)15) + 64;
}

template <typename T>
int launch_top1_union(const T* conf, int P, int N, int M, long long* out, int* count, hipStream_t st, const uint8_t* sm,
                      const uint8_t* tm, void* ws, size_t ws_bytes) {
    if (P <= 0) return DR_OK;
    if (N + M > T1_MAX || (long)N * M >= 0xFFFFFFFFL) return DR_ENOSUP;
    if (!(sm && tm)) sm = tm = nullptr;
    const size_t need = top1_union_workspace_bytes(P, N, M, sizeof(T));
    if (ws && need && ws_bytes >= need) {
        const int NB = (N + T1_ROWS - 1) / T1_ROWS;
        char* w8 = (char*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
        T* colval = (T*)w8; w8 += (size_t)P * NB * M * sizeof(T);
        int* colrow = (int*)w8; w8 += (size_t)P * NB * M * 4;
        unsigned* rowkey = (unsigned*)w8;
        hipLaunchKernelGGL((top1_block_kernel<T>), dim3(NB, P), dim3(1024), 0, st, conf, N, M, rowkey, colval, colrow, sm, tm);
        DR_LAUNCH_CHECK();
        hipLaunchKernelGGL((top1_merge_kernel<T>), dim3(P), dim3(1024), 0, st, N, M, NB, rowkey, colval, colrow, out, count);
        DR_LAUNCH_CHECK();
        return DR_OK;
    }
    hipLaunchKernelGGL((top1_union_kernel<T>), dim3(P), dim3(1024), 0, st, conf, N, M, out, count, sm, tm);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
template int launch_top1_union<float>(const float*, int, int, int, long long*, int*, hipStream_t, const uint8_t*, const uint8_t*, void*, size_t);
// ---------------------------------------------------------------------------------------------
// Matching.get_match(conf, thr, mutual=True) (3D/models/matching.py:126-143; the read-out the 4DMatch tester applies to
// conf_matrix_pred, 4D/lib/tester.py:266): entries that are > thr AND equal to their row maximum AND equal to their column
// maximum (every tie counts, as the reference's `==` does), listed like nonzero(): ascending (b, i, j).  One workgroup per
// pair: column maxima in LDS, then two sweeps over the rows (count per row -> exclusive scan -> ordered write).
// ---------------------------------------------------------------------------------------------
constexpr int MM_MAX = 4096;   // N, M <= MM_MAX

template <typename T>
__global__ __launch_bounds__(1024) void mutual_match_kernel(const T* __restrict__ conf, int N, int M, T thr, int mutual, int cap,
                                                            long long* __restrict__ out, T* __restrict__ mconf, int* __restrict__ count,
                                                            uint8_t* __restrict__ mask) {
    __shared__ T s_col[MM_MAX];
    __shared__ int s_cnt[MM_MAX + 1];
    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const T* c = conf + (size_t)pair * N * M;
    // column maxima: 64 columns at a time, the rows dealt to the 16 waves (lane = column: coalesced, independent loads), the waves' partials
    // combined in wave order.  (One thread per column walking all N rows was a chain of N dependent steps: 190 of this kernel's 207 us at 375 x 381.)
    __shared__ T s_part[16][64];
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        T best = -INFINITY;
        if (mutual && j < M) {
#pragma unroll 4
            for (int i = w; i < N; i += 16) { const T v = c[(size_t)i * M + j]; best = (v > best || v != v) ? v : best; }   // NaN propagates like torch.max
        }
        s_part[w][lane] = best;
        __syncthreads();
        if (w == 0 && j < M) {
            T b = s_part[0][lane];
#pragma unroll
            for (int k = 1; k < 16; ++k) { const T o = s_part[k][lane]; b = (o > b || o != o) ? o : b; }
            s_col[j] = b;
        }
        __syncthreads();
    }
    auto hit = [&](int i, int j, T rowmax) {
        const T v = c[(size_t)i * M + j];
        return v > thr && (!mutual || (v == rowmax && v == s_col[j]));
    };
    auto row_max = [&](int i) {
        T best = -INFINITY;
        if (mutual) {
            for (int j = lane; j < M; j += 64) { const T v = c[(size_t)i * M + j]; best = (v > best || v != v) ? v : best; }
            for (int m = 32; m >= 1; m >>= 1) { const T o = __shfl_xor(best, m); best = (o > best || o != o) ? o : best; }
        }
        return best;
    };
    for (int i = w; i < N; i += 16) {
        const T rm = row_max(i);
        int n = 0;
        for (int j0 = 0; j0 < M; j0 += 64) {
            const int j = j0 + lane;
            n += __popcll(__ballot(j < M && hit(i, j, rm)));
        }
        if (lane == 0) s_cnt[i] = n;
    }
    __syncthreads();
    if (t == 0) {                                    // exclusive scan over the rows (N <= 4096: a few microseconds once per pair)
        int acc = 0;
        for (int i = 0; i < N; ++i) { const int n = s_cnt[i]; s_cnt[i] = acc; acc += n; }
        s_cnt[N] = acc;
        count[pair] = acc;                           // the TRUE count: > cap tells the caller that the list was truncated
    }
    __syncthreads();
    long long* o = out + (size_t)pair * cap * 3;
    for (int i = w; i < N; i += 16) {
        const T rm = row_max(i);
        int base = s_cnt[i];
        for (int j0 = 0; j0 < M; j0 += 64) {
            const int j = j0 + lane;
            const bool h = j < M && hit(i, j, rm);
            const unsigned long long b = __ballot(h);
            if (mask && j < M) mask[((size_t)pair * N + i) * M + j] = h ? 1 : 0;
            if (h) {
                const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
                if (pos < cap) {
                    o[pos * 3] = pair; o[pos * 3 + 1] = i; o[pos * 3 + 2] = j;
                    if (mconf) mconf[(size_t)pair * cap + pos] = c[(size_t)i * M + j];
                }
            }
            base += __popcll(b);
        }
    }
}

template <typename T>
static int launch_mutual_match(const T* conf, int P, int N, int M, T thr, int mutual, int cap, long long* out, T* mconf, int* count,
                               uint8_t* mask, hipStream_t st) {
    if (P <= 0) return DR_OK;
    if (N > MM_MAX || M > MM_MAX) return DR_ENOSUP;
    hipLaunchKernelGGL((mutual_match_kernel<T>), dim3(P), dim3(1024), 0, st, conf, N, M, thr, mutual, cap, out, mconf, count, mask);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
}  // namespace dr
extern "C" {
int dr_mutual_match_f64(int P, int N, int M, const double* conf, double thr, int mutual, int cap, int64_t* matches, double* mconf,
                        int32_t* count, uint8_t* mask, void* stream) {
    if (P < 0 || N < 1 || M < 1 || cap < 1 || !conf || !matches || !count) return DR_EINVAL;
    return dr::launch_mutual_match<double>(conf, P, N, M, thr, mutual, cap, (long long*)matches, mconf, count, mask, (hipStream_t)stream);
}
int dr_mutual_match_f32(int P, int N, int M, const float* conf, float thr, int mutual, int cap, int64_t* matches, float* mconf,
                        int32_t* count, uint8_t* mask, void* stream) {
    if (P < 0 || N < 1 || M < 1 || cap < 1 || !conf || !matches || !count) return DR_EINVAL;
    return dr::launch_mutual_match<float>(conf, P, N, M, thr, mutual, cap, (long long*)matches, mconf, count, mask, (hipStream_t)stream);
}
}
namespace dr {

template int launch_top1_union<double>(const double*, int, int, int, long long*, int*, hipStream_t, const uint8_t*, const uint8_t*, void*, size_t);

}  // namespace dr

// ---------------------------------------------------------------------------------------------
// batch_mutual_topk_select (Diff-Reg-2d3d/vision3d/ops/mutual_topk_select.py:63-134): per batch element the entries that are
// among the k best of their row and / or of their column (mutual: and / or), beyond a threshold, inside the row / column
// masks; returned like torch.nonzero: (b, i, j) in row-major order with their scores.  Used by the fine matching behind the
// 2D-3D loop (EXP/model.py:744-752: k = 2, threshold 0.75) on the patch-to-patch similarity of every node correspondence.
// Equal scores: the lower index wins a place among the k best (torch.topk leaves ties unspecified).
// Three kernels: selection bits + per-batch counts (one workgroup per batch element), scan of the counts, ordered write.
// ---------------------------------------------------------------------------------------------
namespace dr {

constexpr int MT_MAX_K = 8;

struct MtArgs {
    const float* score; const uint8_t* rmask; const uint8_t* cmask; int B, N, M, k, largest, mutual, use_thr; float thr;
    unsigned* bits; int words; int* counts; long long* out_idx; float* out_score; int* total; long long capacity;
};

__device__ __forceinline__ bool mt_before(float v, int i, float pv, int pi, bool largest) {   // (v, i) ranks before (pv, pi)
    return largest ? (v > pv || (v == pv && i < pi)) : (v < pv || (v == pv && i < pi));
}

__global__ __launch_bounds__(256) void mt_select_kernel(MtArgs A) {
    extern __shared__ unsigned mt_bits[];                 // row-side bits, then column-side bits (A.words each)
    unsigned* rb = mt_bits; unsigned* cb = mt_bits + A.words;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* S = A.score + (size_t)b * A.N * A.M;
    const bool largest = A.largest != 0;
    for (int i = t; i < 2 * A.words; i += 256) mt_bits[i] = 0u;
    __syncthreads();
    // rows: one wave per row, k rounds of "best entry after the previous pick"
    for (int i = w; i < A.N; i += 4) {
        float pv = 0.f; int pj = -1;
        for (int r = 0; r < A.k && r < A.M; ++r) {
            float bv = 0.f; int bj = 0x7fffffff;
            for (int j = lane; j < A.M; j += 64) {
                const float v = S[(size_t)i * A.M + j];
                if (v != v) continue;                                     // (NaN never selected)
                if (pj >= 0 && !mt_before(pv, pj, v, j, largest)) continue;   // already picked or ranks before the last pick
                if (bj == 0x7fffffff || mt_before(v, j, bv, bj, largest)) { bv = v; bj = j; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ov = __shfl_xor(bv, m); const int oj = __shfl_xor(bj, m);
                if (oj != 0x7fffffff && (bj == 0x7fffffff || mt_before(ov, oj, bv, bj, largest))) { bv = ov; bj = oj; }
            }
            if (bj == 0x7fffffff) break;
            if (lane == 0) { const int e = i * A.M + bj; atomicOr(&rb[e >> 5], 1u << (e & 31)); }
            pv = bv; pj = bj;
        }
    }
    // columns: one thread per column
    for (int j = t; j < A.M; j += 256) {
        float pv = 0.f; int pi = -1;
        for (int r = 0; r < A.k && r < A.N; ++r) {
            float bv = 0.f; int bi = 0x7fffffff;
            for (int i = 0; i < A.N; ++i) {
                const float v = S[(size_t)i * A.M + j];
                if (v != v) continue;
                if (pi >= 0 && !mt_before(pv, pi, v, i, largest)) continue;
                if (bi == 0x7fffffff || mt_before(v, i, bv, bi, largest)) { bv = v; bi = i; }
            }
            if (bi == 0x7fffffff) break;
            const int e = bi * A.M + j; atomicOr(&cb[e >> 5], 1u << (e & 31));
            pv = bv; pi = bi;
        }
    }
    __syncthreads();
    // combine, threshold, masks -> global bits, count
    int cnt = 0;
    for (int wd = t; wd < A.words; wd += 256) {
        unsigned m = A.mutual ? (rb[wd] & cb[wd]) : (rb[wd] | cb[wd]);
        unsigned keep = 0u;
        while (m) {
            const int bit = __ffs(m) - 1; m &= m - 1;
            const int e = wd * 32 + bit, i = e / A.M, j = e % A.M;
            const float v = S[e];
            bool ok = !A.use_thr || (largest ? v > A.thr : v < A.thr);
            if (A.rmask && !A.rmask[(size_t)b * A.N + i]) ok = false;
            if (A.cmask && !A.cmask[(size_t)b * A.M + j]) ok = false;
            if (ok) keep |= 1u << bit;
        }
        A.bits[(size_t)b * A.words + wd] = keep;
        cnt += __popc(keep);
    }
    __shared__ int s_cnt[4];
    cnt = wave_sum(cnt);
    if (lane == 0) s_cnt[w] = cnt;
    __syncthreads();
    if (t == 0) A.counts[b] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(256) void mt_scan_kernel(int* counts, int B, int* total) {   // counts -> exclusive offsets
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < B ? counts[i] : 0;
        int inc = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(inc, m); if (lane >= m) inc += o; }
        __shared__ int s_w[4];
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int off = s_carry;
        for (int k = 0; k < w; ++k) off += s_w[k];
        if (i < B) counts[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

__global__ __launch_bounds__(256) void mt_write_kernel(MtArgs A) {
    __shared__ int s_w[4];
    __shared__ int s_carry;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* S = A.score + (size_t)b * A.N * A.M;
    if (t == 0) s_carry = A.counts[b];                   // (exclusive offset of this batch element)
    __syncthreads();
    for (int w0 = 0; w0 < A.words; w0 += 256) {
        const int wd = w0 + t;
        unsigned m = wd < A.words ? A.bits[(size_t)b * A.words + wd] : 0u;
        const int c = __popc(m);
        int inc = c;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int off = s_carry;
        for (int k = 0; k < w; ++k) off += s_w[k];
        long long pos = off + inc - c;
        while (m) {
            const int bit = __ffs(m) - 1; m &= m - 1;
            const int e = wd * 32 + bit;
            if (pos < A.capacity) {
                A.out_idx[pos * 3] = b; A.out_idx[pos * 3 + 1] = e / A.M; A.out_idx[pos * 3 + 2] = e % A.M;
                A.out_score[pos] = S[e];
            }
            ++pos;
        }
        __syncthreads();
        if (t == 255) s_carry = off + inc;
        __syncthreads();
    }
}

}  // namespace dr

extern "C" {

size_t dr_mutual_topk_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    const size_t words = ((size_t)N * M + 31) / 32;
    return (size_t)B * words * 4 + (size_t)B * 4 + 256;
}

int dr_mutual_topk_select_f32(int B, int N, int M, const float* score, int k, int largest, int use_threshold, float threshold,
                              int mutual, const uint8_t* row_masks, const uint8_t* col_masks, int64_t* out_idx, float* out_score,
                              long long capacity, int32_t* total, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace dr;
    if (B < 0 || N <= 0 || M <= 0 || k < 1 || k > MT_MAX_K || capacity < 0 || !total) return DR_EINVAL;
    if (k > N || k > M) return DR_EINVAL;                              // torch.topk raises "selected index k out of range" (mutual_topk_select.py:27-28)
    hipStream_t st = (hipStream_t)stream;
    DR_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(int32_t), st));
    if (B == 0) return DR_OK;
    if (!score || (capacity > 0 && (!out_idx || !out_score))) return DR_EINVAL;
    const size_t words = ((size_t)N * M + 31) / 32;
    if (words * 8 > 64 * 1024) return DR_ENOSUP;                       // both bit planes of one element live in LDS
    if (!workspace || workspace_bytes < dr_mutual_topk_workspace_bytes(B, N, M)) return DR_EWORKSPACE;
    MtArgs A{};
    A.score = score; A.rmask = row_masks; A.cmask = col_masks; A.B = B; A.N = N; A.M = M; A.k = k; A.largest = largest; A.mutual = mutual;
    A.use_thr = use_threshold; A.thr = threshold; A.bits = (unsigned*)workspace; A.words = (int)words;
    A.counts = (int*)((char*)workspace + (size_t)B * words * 4); A.out_idx = (long long*)out_idx; A.out_score = out_score; A.total = total;
    A.capacity = capacity;
    hipLaunchKernelGGL(mt_select_kernel, dim3(B), dim3(256), words * 8, st, A);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(mt_scan_kernel, dim3(1), dim3(256), 0, st, A.counts, B, total);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(mt_write_kernel, dim3(B), dim3(256), 0, st, A);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}
