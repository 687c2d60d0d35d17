// collate.hip -- the reference's collate-time native code on device:
//   batched grid subsampling   cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:4-211
//   batched radius neighbours  cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:210-333 (nanoflann kd-tree radiusSearch)
// which build the index arrays the KPFCN backbone consumes (3D/datasets/dataloader.py:13-68, 120-200).  The reference runs
// them on the data-loader's CPU workers; here they are integer / byte work on the device: a 64-bit voxel key per point, one
// bitonic sort of (key, index) pairs per call, segment heads + scan, and for the neighbours a 27-cell sweep over the supports
// sorted by cell.  No host synchronisation, no atomics on floats: every float32 sum runs in the reference's order, so the
// barycentres are bit-exact.
#include "kernels.h"

// float32 arithmetic below restates the C++ expression by expression (the reference is built without FMA contraction)
#pragma clang fp contract(off)

namespace dr {

constexpr int CL_AXIS_BITS = 16;                 // voxel / cell coordinates per axis: 0 .. 65535
constexpr unsigned long long CL_PAD_KEY = ~0ull;

// ------------------------------------------------------------------------------------------------------------
// cloud table: offsets of the stacked clouds, per-cloud origin / first voxel coordinate
// ------------------------------------------------------------------------------------------------------------
struct CloudInfo {
    int begin, end;          // rows of the stacked array
    float ox, oy, oz;        // subsample: origin corner floor(min * (1 / dl)) * dl;  neighbours: min corner
    int lx, ly, lz;          // smallest voxel coordinate of the cloud (keys are stored relative to it)
    int bad;                 // a coordinate range does not fit CL_AXIS_BITS
};

__device__ __forceinline__ int cl_floor_div(float p, float o, float d) { return (int)floorf((p - o) / d); }

// one workgroup per cloud: bounding box -> origin, voxel coordinate range
__global__ __launch_bounds__(1024) void cloud_info_kernel(const float* __restrict__ pts, const int* __restrict__ lengths, int nb,
                                                          float cell, int subsample_origin, CloudInfo* __restrict__ info,
                                                          int* __restrict__ status) {
    __shared__ float s_mn[16][3], s_mx[16][3];
    __shared__ int s_begin;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) {
        int s = 0;
        for (int k = 0; k < b; ++k) s += lengths[k];
        s_begin = s;
    }
    __syncthreads();
    const int begin = s_begin, len = lengths[b];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = t; i < len; i += 1024) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = pts[(size_t)(begin + i) * 3 + c];
            mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    if (lane == 0)
        for (int c = 0; c < 3; ++c) { s_mn[w][c] = mn[c]; s_mx[w][c] = mx[c]; }
    __syncthreads();
    if (t != 0) return;
    for (int k = 1; k < 16; ++k)
        for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], s_mn[k][c]); mx[c] = fmaxf(mx[c], s_mx[k][c]); }
    CloudInfo ci;
    ci.begin = begin; ci.end = begin + len; ci.bad = 0;
    float o[3];
    for (int c = 0; c < 3; ++c) {
        // grid_subsampling.cpp:26  originCorner = floor(minCorner * (1 / sampleDl)) * sampleDl
        o[c] = subsample_origin ? floorf(mn[c] * (1.0f / cell)) * cell : mn[c];
    }
    ci.ox = o[0]; ci.oy = o[1]; ci.oz = o[2];
    int lo[3] = {0, 0, 0};
    if (len > 0) {
        for (int c = 0; c < 3; ++c) {
            lo[c] = cl_floor_div(mn[c], o[c], cell);              // (p - o) / cell is monotone in p: extremes at the corners
            const int hi = cl_floor_div(mx[c], o[c], cell);
            if (hi - lo[c] + 3 >= (1 << CL_AXIS_BITS)) ci.bad = 1;   // (+2: the neighbour sweep looks one cell beyond either end)
        }
    }
    ci.lx = lo[0]; ci.ly = lo[1]; ci.lz = lo[2];
    if (ci.bad || !(cell > 0.f)) atomicExch(status, 1);
    info[b] = ci;
}

__device__ __forceinline__ int cl_cloud_of(const CloudInfo* info, int nb, int row) {
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row >= info[mid].end) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ unsigned long long cl_key(int b, int ix, int iy, int iz) {     // coordinates already offset by +1
    return ((unsigned long long)b << (3 * CL_AXIS_BITS)) | ((unsigned long long)iz << (2 * CL_AXIS_BITS)) |
           ((unsigned long long)iy << CL_AXIS_BITS) | (unsigned long long)ix;
}

// key of every point (rows >= n: padding keys that sort last)
__global__ __launch_bounds__(256) void point_key_kernel(const float* __restrict__ pts, int n, int n_pad, const CloudInfo* __restrict__ info,
                                                        int nb, float cell, unsigned long long* __restrict__ keys,
                                                        unsigned* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    if (i >= n) { keys[i] = CL_PAD_KEY; vals[i] = 0xFFFFFFFFu; return; }
    const int b = cl_cloud_of(info, nb, i);
    const CloudInfo ci = info[b];
    const int ix = cl_floor_div(pts[(size_t)i * 3], ci.ox, cell) - ci.lx + 1;
    const int iy = cl_floor_div(pts[(size_t)i * 3 + 1], ci.oy, cell) - ci.ly + 1;
    const int iz = cl_floor_div(pts[(size_t)i * 3 + 2], ci.oz, cell) - ci.lz + 1;
    const int m = (1 << CL_AXIS_BITS) - 1;
    keys[i] = cl_key(b, min(max(ix, 0), m), min(max(iy, 0), m), min(max(iz, 0), m));
    vals[i] = (unsigned)i;
}

// ------------------------------------------------------------------------------------------------------------
// bitonic sort of (key, value) pairs, ascending by (key, value): n_pad a power of two.  Chunks of 2048 pairs are sorted /
// merged in LDS (strides <= 1024); strides >= 2048 are one global compare-exchange pass each.
// ------------------------------------------------------------------------------------------------------------
constexpr int BS_CHUNK = 2048;

__device__ __forceinline__ bool bs_greater(unsigned long long ka, unsigned va, unsigned long long kb, unsigned vb) {
    return ka > kb || (ka == kb && va > vb);
}

// k_first .. k_last: the merge sizes handled inside the chunk; for every k, strides j = min(k / 2, 1024) .. 1
__global__ __launch_bounds__(1024) void bitonic_local_kernel(unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, int n_pad,
                                                             int k_first, int k_last) {
    __shared__ unsigned long long s_k[BS_CHUNK];
    __shared__ unsigned s_v[BS_CHUNK];
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * BS_CHUNK;
    const int cnt = min(BS_CHUNK, n_pad - (int)base);        // n_pad < 2048: one short chunk
    for (int i = t; i < cnt; i += 1024) { s_k[i] = keys[base + i]; s_v[i] = vals[base + i]; }
    __syncthreads();
    for (int k = k_first; k <= k_last; k <<= 1) {
        for (int j = min(k >> 1, BS_CHUNK / 2); j > 0; j >>= 1) {
            for (int p = t; p < cnt / 2; p += 1024) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));          // lower index of pair p at stride j
                const int l = i | j;
                const bool up = (((base + i) & (size_t)k) == 0);
                const unsigned long long ka = s_k[i], kb = s_k[l];
                const unsigned va = s_v[i], vb = s_v[l];
                if (bs_greater(ka, va, kb, vb) == up) { s_k[i] = kb; s_k[l] = ka; s_v[i] = vb; s_v[l] = va; }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < cnt; i += 1024) { keys[base + i] = s_k[i]; vals[base + i] = s_v[i]; }
}

__global__ __launch_bounds__(256) void bitonic_global_kernel(unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, int n_pad,
                                                             int k, int j) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)n_pad / 2) return;
    const size_t i = ((p & ~(size_t)(j - 1)) << 1) | (p & (size_t)(j - 1));
    const size_t l = i | (size_t)j;
    const bool up = ((i & (size_t)k) == 0);
    const unsigned long long ka = keys[i], kb = keys[l];
    const unsigned va = vals[i], vb = vals[l];
    if (bs_greater(ka, va, kb, vb) == up) { keys[i] = kb; keys[l] = ka; vals[i] = vb; vals[l] = va; }
}

static int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

int launch_bitonic_sort(unsigned long long* keys, unsigned* vals, int n_pad, hipStream_t st) {   // (kernels.h: also sorts the keys of fine2d3d.hip)
    const int chunks = (n_pad + BS_CHUNK - 1) / BS_CHUNK;
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(chunks), dim3(1024), 0, st, keys, vals, n_pad, 2, min(n_pad, BS_CHUNK));
    DR_LAUNCH_CHECK();
    for (int k = 2 * BS_CHUNK; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j >= BS_CHUNK; j >>= 1) {
            hipLaunchKernelGGL(bitonic_global_kernel, dim3((n_pad / 2 + 255) / 256), dim3(256), 0, st, keys, vals, n_pad, k, j);
            DR_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(bitonic_local_kernel, dim3(chunks), dim3(1024), 0, st, keys, vals, n_pad, k, k);
        DR_LAUNCH_CHECK();
    }
    return DR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// segment heads of the sorted keys -> rank of every voxel (three-kernel exclusive scan of the head flags)
// ------------------------------------------------------------------------------------------------------------
constexpr int SC_BLOCK = 1024, SC_PER = 4;      // 4096 flags per workgroup

__device__ __forceinline__ int cl_is_head(const unsigned long long* keys, int i, int n) {
    return i < n && (i == 0 || keys[i] != keys[i - 1]);
}
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_up(inc, m);
        if (lane >= m) inc += o;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < 16; ++k) { const int x = s_w[k]; if (k < w) off += x; tot += x; }
    __syncthreads();
    total = tot;
    return off + inc - v;
}
__global__ __launch_bounds__(SC_BLOCK) void heads_count_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ block_sum) {
    __shared__ int s_w[16];
    const int base = blockIdx.x * SC_BLOCK * SC_PER + threadIdx.x * SC_PER;
    int c = 0;
#pragma unroll
    for (int e = 0; e < SC_PER; ++e) c += cl_is_head(keys, base + e, n);
    int tot;
    block_excl_scan_1024(c, s_w, tot);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SC_BLOCK) void heads_blockscan_kernel(int* __restrict__ block_sum, int nblk, int* __restrict__ total) {
    __shared__ int s_w[16];
    int carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += SC_BLOCK) {
        const int i = b0 + threadIdx.x;
        const int v = i < nblk ? block_sum[i] : 0;
        int tot;
        const int ex = block_excl_scan_1024(v, s_w, tot);
        if (i < nblk) block_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

// barycentre of every voxel: the thread of a segment head walks its segment; the (key, index) sort left the points of a
// voxel in input order, so the float32 sum is the reference's (grid_subsampling.cpp:59-66: point += p per input point)
__global__ __launch_bounds__(SC_BLOCK) void barycentre_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                              int n, const int* __restrict__ block_off, const float* __restrict__ pts,
                                                              float* __restrict__ out, int* __restrict__ out_lengths) {
    __shared__ int s_w[16];
    const int base = blockIdx.x * SC_BLOCK * SC_PER + threadIdx.x * SC_PER;
    int flags[SC_PER], c = 0;
#pragma unroll
    for (int e = 0; e < SC_PER; ++e) { flags[e] = cl_is_head(keys, base + e, n); c += flags[e]; }
    int tot;
    int rank = block_off[blockIdx.x] + block_excl_scan_1024(c, s_w, tot);
#pragma unroll
    for (int e = 0; e < SC_PER; ++e) {
        if (!flags[e]) continue;
        const int i = base + e;
        const unsigned long long k = keys[i];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        int cnt = 0;
        for (int j = i; j < n && keys[j] == k; ++j) {
            const float* p = pts + (size_t)vals[j] * 3;
            sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2];
            ++cnt;
        }
        const float inv = (float)(1.0 / (double)cnt);       // grid_subsampling.cpp:88  point * (1.0 / count)
        out[(size_t)rank * 3] = sx * inv; out[(size_t)rank * 3 + 1] = sy * inv; out[(size_t)rank * 3 + 2] = sz * inv;
        atomicAdd(out_lengths + (int)(k >> (3 * CL_AXIS_BITS)), 1);
        ++rank;
    }
}

// ------------------------------------------------------------------------------------------------------------
// radius neighbours: supports sorted by cell (cell edge = radius * (1 + 2^-7): two points closer than the radius are then at
// most one cell apart on every axis whatever the rounding of the float32 cell coordinate), 3 x 3 runs of x-adjacent cells per
// query, the `limit` nearest kept sorted in LDS (one list per thread, interleaved over the 64 lanes)
// ------------------------------------------------------------------------------------------------------------
constexpr int NB_MAX_LIMIT = 64;

__global__ __launch_bounds__(256) void gather_sorted_kernel(const unsigned* __restrict__ vals, int n, const float* __restrict__ pts,
                                                            float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned v = vals[i];
    sorted[i] = make_float4(pts[(size_t)v * 3], pts[(size_t)v * 3 + 1], pts[(size_t)v * 3 + 2], __uint_as_float(v));
}

// cell -> (first sorted position, number of points): open-addressing hash table filled from the segment heads of the sorted
// keys.  A query then finds its 27 cells with 27 independent probes instead of dependent binary searches (18 x 17 loads in a
// row made the first version latency-bound: 0.65 ms for 57 k queries at 5 waves per CU).
struct CellSlot { unsigned long long key; int start, len; };

__device__ __forceinline__ unsigned cl_hash(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)(z ^ (z >> 31));
}

__global__ __launch_bounds__(256) void cell_table_kernel(const unsigned long long* __restrict__ keys, int n, CellSlot* __restrict__ tab,
                                                         unsigned mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (!cl_is_head(keys, i, n)) return;
    const unsigned long long k = keys[i];
    int j = i + 1;
    while (j < n && keys[j] == k) ++j;
    unsigned slot = cl_hash(k) & mask;
    while (true) {
        const unsigned long long prev = atomicCAS(&tab[slot].key, CL_PAD_KEY, k);
        if (prev == CL_PAD_KEY) break;                 // (keys of heads are distinct: no duplicate inserts)
        slot = (slot + 1) & mask;
    }
    tab[slot].start = i;
    tab[slot].len = j - i;
}

struct NbArgs {
    const float* queries; int nq; const CloudInfo* q_info; const CloudInfo* s_info; int nb;
    const CellSlot* tab; unsigned mask; const float4* sorted; int ns;
    float cell, r2; int limit; long long* out; int* max_count;
};

constexpr int NB_CAP = 64;              // candidates kept per query before the final ranking (>= limit)
constexpr int NB_STRIDE = NB_CAP + 1;   // list stride in LDS (odd: the lanes' append positions fall into different banks)

__global__ __launch_bounds__(64) void radius_query_kernel(NbArgs A) {
    // one list per lane, list-major: entry e of lane L at [L * NB_STRIDE + e]
    __shared__ float s_d[64 * NB_STRIDE];
    __shared__ unsigned s_i[64 * NB_STRIDE];
    const int lane = threadIdx.x, q = blockIdx.x * 64 + lane;
    int cnt = 0, kept = 0;
    float* my_d = s_d + lane * NB_STRIDE;
    unsigned* my_i = s_i + lane * NB_STRIDE;
    if (q < A.nq) {
        const int b = cl_cloud_of(A.q_info, A.nb, q);
        const CloudInfo si = A.s_info[b];
        const float qx = A.queries[(size_t)q * 3], qy = A.queries[(size_t)q * 3 + 1], qz = A.queries[(size_t)q * 3 + 2];
        if (si.end > si.begin) {
            const int m = (1 << CL_AXIS_BITS) - 1;
            // cell of the query in the support cloud's grid (may lie outside it)
            const long long cx = (long long)floorf((qx - si.ox) / A.cell) - si.lx + 1;
            const long long cy = (long long)floorf((qy - si.oy) / A.cell) - si.ly + 1;
            const long long cz = (long long)floorf((qz - si.oz) / A.cell) - si.lz + 1;
            // per (y, z) row: three independent probes (x - 1, x, x + 1 are adjacent keys, so the cells that exist form ONE
            // run of the sorted supports), then the sweep of that run
#pragma unroll 1
            for (int row = 0; row < 9; ++row) {
                const long long y = cy + (row % 3) - 1, z = cz + (row / 3) - 1;
                if (y < 0 || y > m || z < 0 || z > m) continue;
                int lo = 0x7fffffff, hi = 0;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const long long x = cx + dx;
                    if (x < 0 || x > m) continue;
                    const unsigned long long k = cl_key(b, (int)x, (int)y, (int)z);
                    unsigned slot = cl_hash(k) & A.mask;
                    while (true) {
                        const unsigned long long tk = A.tab[slot].key;
                        if (tk == k) {
                            const int s0 = A.tab[slot].start;
                            lo = min(lo, s0); hi = max(hi, s0 + A.tab[slot].len);
                            break;
                        }
                        if (tk == CL_PAD_KEY) break;
                        slot = (slot + 1) & A.mask;
                    }
                }
                // candidates eight at a time: the loads of a batch are independent and in flight together
                for (int j0 = lo; j0 < hi; j0 += 8) {
                    float4 sv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sv[u] = A.sorted[min(j0 + u, hi - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (j0 + u >= hi) break;
                        const float4 s = sv[u];
                        // nanoflann L2_Simple_Adaptor: result += diff * diff over the three axes, float
                        const float d0 = qx - s.x, d1 = qy - s.y, d2 = qz - s.z;
                        const float dd = ((0.f + d0 * d0) + d1 * d1) + d2 * d2;
                        if (!(dd < A.r2)) continue;
                        ++cnt;
                        const unsigned id = __float_as_uint(s.w);
                        // APPEND, unsorted (sorting on arrival made every candidate step of the wave pay the longest
                        // shift of any lane: 530 us of a 590 us launch); the lists are ranked once, cooperatively, below
                        if (kept < NB_CAP) {
                            my_d[kept] = dd; my_i[kept] = id;
                            ++kept;
                        } else {          // more than NB_CAP points in the ball: the farthest kept one makes room
                            int far = 0;
                            float fd = my_d[0]; unsigned fi = my_i[0];
                            for (int e = 1; e < NB_CAP; ++e) {
                                const float ed = my_d[e]; const unsigned ei = my_i[e];
                                if (ed > fd || (ed == fd && ei > fi)) { far = e; fd = ed; fi = ei; }
                            }
                            if (dd < fd || (dd == fd && id < fi)) { my_d[far] = dd; my_i[far] = id; }
                        }
                    }
                }
            }
        }
    }
    const int mx = wave_max(cnt);
    if (lane == 0 && mx > 0) atomicMax(A.max_count, mx);
    __syncthreads();
    // rank the 64 lists one after the other with the whole wave: lane i owns entry i of list L and counts the entries that
    // precede it in (distance, index) order (broadcast reads); entry -> column `rank` of the row, columns >= n padded
    for (int L = 0; L < 64; ++L) {
        const int qL = blockIdx.x * 64 + L;
        if (qL >= A.nq) break;
        const int n = __shfl(kept, L);
        const float* ld = s_d + L * NB_STRIDE;
        const unsigned* li = s_i + L * NB_STRIDE;
        long long* o = A.out + (size_t)qL * A.limit;
        if (lane < n) {
            const float di = ld[lane]; const unsigned ii = li[lane];
            int rank = 0;
            for (int j = 0; j < n; j += 8) {           // eight broadcast reads in flight (one at a time = one LDS round trip each)
                float dj[8]; unsigned ij[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { dj[u] = ld[min(j + u, n - 1)]; ij[u] = li[min(j + u, n - 1)]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += (j + u < n && (dj[u] < di || (dj[u] == di && ij[u] < ii))) ? 1 : 0;
            }
            if (rank < A.limit) o[rank] = (long long)ii;
        }
        if (lane >= n && lane < A.limit) o[lane] = (long long)A.ns;                 // pad: neighbors.cpp:325
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace dr

using namespace dr;

extern "C" {

size_t dr_grid_subsample_workspace_bytes(int n, int nb) {
    if (n <= 0 || nb <= 0) return 0;
    const int n_pad = next_pow2(n);
    const int nblk = (n + SC_BLOCK * SC_PER - 1) / (SC_BLOCK * SC_PER);
    return align256(sizeof(CloudInfo) * nb) + align256(8ull * n_pad) + align256(4ull * n_pad) + align256(4ull * nblk);
}

int dr_grid_subsample_f32(int n, int nb, const float* points, const int32_t* lengths, float dl, float* out_points,
                          int32_t* out_lengths, int32_t* out_total, int32_t* status, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (n < 0 || nb <= 0 || nb >= (1 << 15) || !(dl > 0.f)) return DR_EINVAL;
    if (!lengths || !out_lengths || !out_total || !status) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DR_HIP_CHECK(hipMemsetAsync(out_lengths, 0, sizeof(int32_t) * nb, st));
    DR_HIP_CHECK(hipMemsetAsync(out_total, 0, sizeof(int32_t), st));
    DR_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    if (n == 0) return DR_OK;
    if (!points || !out_points) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_grid_subsample_workspace_bytes(n, nb)) return DR_EWORKSPACE;
    const int n_pad = next_pow2(n);
    const int nblk = (n + SC_BLOCK * SC_PER - 1) / (SC_BLOCK * SC_PER);
    char* w = (char*)workspace;
    CloudInfo* info = (CloudInfo*)w; w += align256(sizeof(CloudInfo) * nb);
    unsigned long long* keys = (unsigned long long*)w; w += align256(8ull * n_pad);
    unsigned* vals = (unsigned*)w; w += align256(4ull * n_pad);
    int* block_off = (int*)w;
    hipLaunchKernelGGL(cloud_info_kernel, dim3(nb), dim3(1024), 0, st, points, lengths, nb, dl, 1, info, status);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(point_key_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, points, n, n_pad, info, nb, dl, keys, vals);
    DR_LAUNCH_CHECK();
    int rc = launch_bitonic_sort(keys, vals, n_pad, st);
    if (rc != DR_OK) return rc;
    hipLaunchKernelGGL(heads_count_kernel, dim3(nblk), dim3(SC_BLOCK), 0, st, keys, n, block_off);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(heads_blockscan_kernel, dim3(1), dim3(SC_BLOCK), 0, st, block_off, nblk, out_total);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(barycentre_kernel, dim3(nblk), dim3(SC_BLOCK), 0, st, keys, vals, n, block_off, points, out_points, out_lengths);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

size_t dr_radius_neighbors_workspace_bytes(int nq, int ns, int nb) {
    if (ns <= 0 || nb <= 0) return 0;
    const int n_pad = next_pow2(ns);
    return 2 * align256(sizeof(CloudInfo) * nb) + align256(8ull * n_pad) + align256(4ull * n_pad) + align256(16ull * ns) +
           align256(sizeof(CellSlot) * 2ull * n_pad);
}

int dr_radius_neighbors_f32(int nq, int ns, int nb, const float* queries, const float* supports, const int32_t* q_lengths,
                            const int32_t* s_lengths, float radius, int limit, int64_t* out, int32_t* max_count, int32_t* status,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (nq < 0 || ns < 0 || nb <= 0 || nb >= (1 << 15) || !(radius > 0.f) || limit < 1 || limit > NB_MAX_LIMIT) return DR_EINVAL;
    if (!q_lengths || !s_lengths || !max_count || !status) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DR_HIP_CHECK(hipMemsetAsync(max_count, 0, sizeof(int32_t), st));
    DR_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    if (nq == 0) return DR_OK;
    if (!queries || !out || (ns > 0 && !supports)) return DR_EINVAL;
    if (ns > 0 && (!workspace || workspace_bytes < dr_radius_neighbors_workspace_bytes(nq, ns, nb))) return DR_EWORKSPACE;
    const int n_pad = next_pow2(ns > 0 ? ns : 1);
    char* w = (char*)workspace;
    CloudInfo* q_info = (CloudInfo*)w; w += align256(sizeof(CloudInfo) * nb);
    CloudInfo* s_info = (CloudInfo*)w; w += align256(sizeof(CloudInfo) * nb);
    unsigned long long* keys = (unsigned long long*)w; w += align256(8ull * n_pad);
    unsigned* vals = (unsigned*)w; w += align256(4ull * n_pad);
    float4* sorted = (float4*)w; w += align256(16ull * (ns > 0 ? ns : 1));
    CellSlot* tab = (CellSlot*)w;
    const unsigned tab_size = 2u * (unsigned)n_pad;
    const float cell = radius * (1.0f + 1.0f / 128.0f);
    NbArgs A{};
    A.queries = queries; A.nq = nq; A.q_info = q_info; A.s_info = s_info; A.nb = nb; A.tab = tab; A.mask = tab_size - 1; A.sorted = sorted; A.ns = ns;
    A.cell = cell; A.r2 = radius * radius; A.limit = limit; A.out = (long long*)out; A.max_count = max_count;
    if (ns > 0) {
        // (the query table only provides the cloud boundaries of the stacked queries)
        hipLaunchKernelGGL(cloud_info_kernel, dim3(nb), dim3(1024), 0, st, queries, q_lengths, nb, cell, 0, q_info, status);
        DR_LAUNCH_CHECK();
        hipLaunchKernelGGL(cloud_info_kernel, dim3(nb), dim3(1024), 0, st, supports, s_lengths, nb, cell, 0, s_info, status);
        DR_LAUNCH_CHECK();
        hipLaunchKernelGGL(point_key_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, supports, ns, n_pad, s_info, nb, cell, keys, vals);
        DR_LAUNCH_CHECK();
        int rc = launch_bitonic_sort(keys, vals, n_pad, st);
        if (rc != DR_OK) return rc;
        hipLaunchKernelGGL(gather_sorted_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, vals, ns, supports, sorted);
        DR_LAUNCH_CHECK();
        DR_HIP_CHECK(hipMemsetAsync(tab, 0xFF, sizeof(CellSlot) * (size_t)tab_size, st));
        hipLaunchKernelGGL(cell_table_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, keys, ns, tab, tab_size - 1);
        DR_LAUNCH_CHECK();
        hipLaunchKernelGGL(radius_query_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, A);
        DR_LAUNCH_CHECK();
    }
    return DR_OK;
}

}  // extern "C"
