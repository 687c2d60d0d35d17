// metrics.hip -- the evaluation harness that consumes the loop's match_pred, on device:
//   inlier ratio / FMR        MatchMotionLoss.compute_inlier_ratio            3D/models/loss.py:383-410
//   NR-FMR                    compute_nrfmr + blend_anchor_motion             3D/lib/tester.py:127-210
//   correspondence RANSAC     ransac_regist_coarse -> Open3D 0.13 RANSAC      3D/models/loss.py:13-24, 347-379
//   registration recall       compute_registration_recall                     3D/models/loss.py:27-44, 415-448
// The reference does all four on the host per pair (device -> host copies of the clouds and matches, Open3D on
// CPU threads); here a batch of P pairs stays on the device and nothing synchronises.
//
// Arithmetic follows the reference: float32 where it computes in float32 (IR, NR-FMR: torch / numpy float32,
// un-fused squares and sums), float64 where it computes in float64 (Open3D points are Vector3d, numpy 4x4 poses).
#include "kernels.h"
#include "svd3.h"

// The float32 metrics restate torch / numpy expressions operation by operation: no implicit fusing of a*b + c in this
// file (contraction is switched off for everything below); every fused
// multiply-add below is written as fma() / fmaf().
#pragma clang fp contract(off)

namespace dr {

// plain float32 operations compiled under the pragma above (the header's __fmul_rn / __fadd_rn bodies are compiled
// under the header's own contraction state and do get fused)
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }

// ------------------------------------------------------------------------------------------------------------
// counter-based sampling: the integer stream of diffreg_hip.synth.hash_bits (splitmix64 finaliser)
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t stream_base(uint64_t seed, uint64_t stream) {
    return mix64(seed * 0x100000001B3ull + stream);
}
__device__ __forceinline__ uint64_t hash_bits(uint64_t base, uint64_t idx) { return mix64(mix64(idx ^ base) + base) >> 11; }

// R x + t the way a float32 matmul accumulates it (k ascending, fused), then the translation added
__device__ __forceinline__ void warp_f32(const float* __restrict__ R, const float* __restrict__ t, float x, float y, float z,
                                         float& ox, float& oy, float& oz) {
    ox = add_rn(fmaf(R[2], z, fmaf(R[1], y, mul_rn(R[0], x))), t[0]);
    oy = add_rn(fmaf(R[5], z, fmaf(R[4], y, mul_rn(R[3], x))), t[1]);
    oz = add_rn(fmaf(R[8], z, fmaf(R[7], y, mul_rn(R[6], x))), t[2]);
}
// sum of three squares as torch.sum / np.sum over an axis of length 3 form it: products first, then left to right
__device__ __forceinline__ float sq3(float a, float b, float c) {
    return add_rn(add_rn(mul_rn(a, a), mul_rn(b, b)), mul_rn(c, c));
}

// ------------------------------------------------------------------------------------------------------------
// inlier ratio
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inlier_kernel(const long long* __restrict__ matches, const int* __restrict__ count,
                                                     int cap, int N, int M, const float* __restrict__ s_pcd,
                                                     const float* __restrict__ t_pcd, const float* __restrict__ rot,
                                                     const float* __restrict__ trn, const float* __restrict__ flow,
                                                     float thr2, int* __restrict__ n_inl) {
    const int pair = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    bool inl = false;
    if (k < count[pair] && k < cap) {
        const long long* m = matches + ((size_t)pair * cap + k) * 3;
        const long long i = m[1], j = m[2];
        if ((unsigned long long)i < (unsigned long long)N && (unsigned long long)j < (unsigned long long)M) {
            const float* s = s_pcd + ((size_t)pair * N + i) * 3;
            const float* y = t_pcd + ((size_t)pair * M + j) * 3;
            float x0 = s[0], x1 = s[1], x2 = s[2];
            if (flow) {   // s_pcd + s2t_flow (loss.py:389)
                const float* f = flow + ((size_t)pair * N + i) * 3;
                x0 = add_rn(x0, f[0]); x1 = add_rn(x1, f[1]); x2 = add_rn(x2, f[2]);
            }
            float wx, wy, wz;
            warp_f32(rot + (size_t)pair * 9, trn + (size_t)pair * 3, x0, x1, x2, wx, wy, wz);
            inl = sq3(sub_rn(wx, y[0]), sub_rn(wy, y[1]), sub_rn(wz, y[2])) < thr2;
        }
    }
    const unsigned long long b = __ballot(inl);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_inl + pair, __popcll(b));
}

__global__ void ratio_kernel(int P, const int* __restrict__ num, const int* __restrict__ den, const int* __restrict__ den_off,
                             int min_den, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int d = den_off ? den_off[p + 1] - den_off[p] : den[p];
    out[p] = d < min_den ? 0.f : (float)num[p] / (float)d;
}

// ------------------------------------------------------------------------------------------------------------
// NR-FMR: 3 nearest motion anchors per metric point, inverse-distance blend
// ------------------------------------------------------------------------------------------------------------
constexpr int NR_TILE = 1024;

struct NrArgs {
    const long long* matches; const int* count; int cap, N, M;
    const float* s_pcd; const float* t_pcd; const float* raw_pcd; const float* raw_flow; const int* raw_off;
    const long long* metric_index; const int* q_off; const float* rot; const float* trn;
    float radius, thr; int* n_hit; float* blended;
};

__global__ __launch_bounds__(256) void nrfmr_kernel(NrArgs A) {
    __shared__ float s_a[NR_TILE * 3];
    const int pair = blockIdx.y, t = threadIdx.x;
    const int K = min(A.count[pair], A.cap);
    const int q0 = A.q_off[pair], Q = A.q_off[pair + 1] - q0;
    if (blockIdx.x * 256 >= Q || K < 4) return;       // np.argpartition(kth=3) needs 4 anchors (datasets/utils.py:13)
    const int q = blockIdx.x * 256 + t;
    const bool active = q < Q;
    const float* s_pcd = A.s_pcd + (size_t)pair * A.N * 3;
    const float* t_pcd = A.t_pcd + (size_t)pair * A.M * 3;
    const long long* mt = A.matches + (size_t)pair * A.cap * 3;
    float px = 0, py = 0, pz = 0, fx = 0, fy = 0, fz = 0;
    if (active) {
        const size_t r = (size_t)A.raw_off[pair] + (size_t)A.metric_index[q0 + q];
        px = A.raw_pcd[r * 3]; py = A.raw_pcd[r * 3 + 1]; pz = A.raw_pcd[r * 3 + 2];
        fx = A.raw_flow[r * 3]; fy = A.raw_flow[r * 3 + 1]; fz = A.raw_flow[r * 3 + 2];
    }
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int k0 = 0; k0 < K; k0 += NR_TILE) {
        const int nk = min(NR_TILE, K - k0);
        for (int k = t; k < nk; k += 256) {
            const long long i = mt[(size_t)(k0 + k) * 3 + 1];
            const bool okk = (unsigned long long)i < (unsigned long long)A.N;
            s_a[k * 3] = okk ? s_pcd[i * 3] : INFINITY;
            s_a[k * 3 + 1] = okk ? s_pcd[i * 3 + 1] : INFINITY;
            s_a[k * 3 + 2] = okk ? s_pcd[i * 3 + 2] : INFINITY;
        }
        __syncthreads();
        if (active) {
            for (int k = 0; k < nk; ++k) {
                const float d = sq3(sub_rn(s_a[k * 3], px), sub_rn(s_a[k * 3 + 1], py), sub_rn(s_a[k * 3 + 2], pz));
                if (d < d2) {       // ascending insertion; equal distances keep the earlier anchor first
                    if (d < d1) {
                        d2 = d1; i2 = i1;
                        if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = k0 + k; }
                        else { d1 = d; i1 = k0 + k; }
                    } else { d2 = d; i2 = k0 + k; }
                }
            }
        }
        __syncthreads();
    }
    bool hit = false;
    if (active) {
        float dd[3] = {sqrtf(d0), sqrtf(d1), sqrtf(d2)};
        const int id[3] = {i0, i1, i2};
        float w[3], mo[3][3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            if (dd[n] < 1e-10f) dd[n] = 1e-10f;             // tester.py:139
            if (dd[n] > A.radius) dd[n] = 1e10f;            // tester.py:140-141
            w[n] = 1.0f / dd[n];
            const long long i = mt[(size_t)id[n] * 3 + 1], j = mt[(size_t)id[n] * 3 + 2];
            const bool okj = (unsigned long long)j < (unsigned long long)A.M;
#pragma unroll
            for (int c = 0; c < 3; ++c) mo[n][c] = okj ? sub_rn(t_pcd[j * 3 + c], s_pcd[i * 3 + c]) : 0.f;   // tester.py:169
        }
        const float ws = add_rn(add_rn(w[0], w[1]), w[2]);
#pragma unroll
        for (int n = 0; n < 3; ++n) w[n] = w[n] / ws;
        float bl[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            bl[c] = add_rn(add_rn(mul_rn(mo[0][c], w[0]), mul_rn(mo[1][c], w[1])), mul_rn(mo[2][c], w[2]));
        if (A.blended) {
            float* o = A.blended + (size_t)(q0 + q) * 3;
            o[0] = bl[0]; o[1] = bl[1]; o[2] = bl[2];
        }
        float gx, gy, gz;
        warp_f32(A.rot + (size_t)pair * 9, A.trn + (size_t)pair * 3, add_rn(px, fx), add_rn(py, fy), add_rn(pz, fz), gx, gy,
                 gz);
        const float e = sqrtf(sq3(sub_rn(add_rn(px, bl[0]), gx), sub_rn(add_rn(py, bl[1]), gy),
                                  sub_rn(add_rn(pz, bl[2]), gz)));
        hit = e < A.thr;
    }
    const unsigned long long b = __ballot(hit);
    if ((t & 63) == 0 && b) atomicAdd(A.n_hit + pair, __popcll(b));
}

// ------------------------------------------------------------------------------------------------------------
// correspondence RANSAC
// ------------------------------------------------------------------------------------------------------------
constexpr int RS_BLOCK = 256;    // hypotheses per workgroup

struct RsBest { double err; int cnt; int it; };
__device__ __forceinline__ bool rs_better(int c, double e, int it, int oc, double oe, int oit) {
    return c > oc || (c == oc && (e < oe || (e == oe && it < oit)));
}

struct RsArgs {
    const long long* matches; const int* count; int cap, N, M;
    const float* s_pcd; const float* t_pcd; const long long* pair_ids;
    double thr2; int iters; uint64_t seed; RsBest* blk; int nblk; double* pts;
    double* rot; double* trn; double* fitness; double* rmse; int* best_iter;
};

// rigid fit of the three correspondences of hypothesis `it` (Eigen::umeyama without scaling:
// sigma = 1/3 sum (y - ybar)(x - xbar)^T = U D V^T, R = U diag(1,1,det U det V) V^T, t = ybar - R xbar)
__device__ __forceinline__ bool rs_fit(const long long* __restrict__ mt, int K, uint64_t base, int it, const float* __restrict__ sp,
                                       const float* __restrict__ tp, int N, int M, double (&R)[3][3], double (&tv)[3]) {
    long long si[3], tj[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int c = (int)(hash_bits(base, (uint64_t)it * 3 + s) % (uint64_t)K);
        si[s] = mt[(size_t)c * 3 + 1];
        tj[s] = mt[(size_t)c * 3 + 2];
        if ((unsigned long long)si[s] >= (unsigned long long)N || (unsigned long long)tj[s] >= (unsigned long long)M) return false;
    }
    if (si[0] == si[1] || si[0] == si[2] || si[1] == si[2] || tj[0] == tj[1] || tj[0] == tj[2] || tj[1] == tj[2]) return false;
    double x[3][3], y[3][3], mx[3], my[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[s][c] = (double)sp[si[s] * 3 + c];
            y[s][c] = (double)tp[tj[s] * 3 + c];
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mx[c] = (x[0][c] + x[1][c] + x[2][c]) * (1.0 / 3.0);
        my[c] = (y[0][c] + y[1][c] + y[2][c]) * (1.0 / 3.0);
    }
    double Sg[3][3], big = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < 3; ++s) v += (y[s][a] - my[a]) * (x[s][b] - mx[b]);
            Sg[a][b] = v;
            big = fmax(big, fabs(v));
        }
    if (!(big > 0.0)) return false;
    const double inv = 1.0 / big;        // O(1) entries: the Jacobi floor below is then scale-free
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Sg[a][b] *= inv;
    double U[3][3], V[3][3], D[3];
    svd3_jacobi(Sg, U, D, V, 1e-56);
    const double dd = det3(U) * det3(V) < 0.0 ? -1.0 : 1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) R[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + dd * U[a][2] * V[b][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) tv[a] = my[a] - (R[a][0] * mx[0] + R[a][1] * mx[1] + R[a][2] * mx[2]);
    return true;
}

// the correspondences of every pair as fp64 point pairs [P, cap, 6] (x, y, z of the source, then of the target; NaN for
// indices outside the clouds): written once, then read by every hypothesis of the pair
__global__ __launch_bounds__(256) void ransac_gather_kernel(RsArgs A) {
    const int pair = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= A.cap) return;
    const long long* m = A.matches + ((size_t)pair * A.cap + k) * 3;
    double* o = A.pts + ((size_t)pair * A.cap + k) * 6;
    const long long i = m[1], j = m[2];
    const bool okk = k < A.count[pair] && (unsigned long long)i < (unsigned long long)A.N && (unsigned long long)j < (unsigned long long)A.M;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = okk ? (double)A.s_pcd[((size_t)pair * A.N + i) * 3 + c] : (double)NAN;
        o[3 + c] = okk ? (double)A.t_pcd[((size_t)pair * A.M + j) * 3 + c] : (double)NAN;
    }
}

__global__ __launch_bounds__(RS_BLOCK) void ransac_eval_kernel(RsArgs A) {
    __shared__ RsBest s_best[RS_BLOCK / 64];
    const int pair = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int K = min(A.count[pair], A.cap);
    RsBest* out = A.blk + (size_t)pair * A.nblk + blockIdx.x;
    if (K < 3) {
        if (t == 0) { out->err = 0.0; out->cnt = -1; out->it = 0x7fffffff; }
        return;
    }
    const float* sp = A.s_pcd + (size_t)pair * A.N * 3;
    const float* tp = A.t_pcd + (size_t)pair * A.M * 3;
    const long long* mt = A.matches + (size_t)pair * A.cap * 3;
    const uint64_t base = stream_base(A.seed, A.pair_ids ? (uint64_t)A.pair_ids[pair] : (uint64_t)pair);
    const int it = blockIdx.x * RS_BLOCK + t;
    double R[3][3], tv[3];
    const bool valid = it < A.iters && rs_fit(mt, K, base, it, sp, tp, A.N, A.M, R, tv);
    if (!valid) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            tv[a] = 0.0;
#pragma unroll
            for (int b = 0; b < 3; ++b) R[a][b] = 0.0;
        }
    }
    int cnt = 0;
    double err = 0.0;
    // every lane scores the SAME correspondence at the same time: the point pair is wave-uniform, so it comes through the
    // scalar cache into SGPRs (s_load_dwordx4) and feeds the FMAs as their scalar operand -- no LDS staging, no barriers
    // (the LDS-broadcast form spent 3 ds_read_b128 per evaluation and kept the fp64 pipe at half rate)
    const double* __restrict__ cp = A.pts + (size_t)pair * A.cap * 6;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        const double sx = cp[k * 6], sy = cp[k * 6 + 1], sz = cp[k * 6 + 2];
        const double ex = fma(R[0][0], sx, fma(R[0][1], sy, fma(R[0][2], sz, tv[0]))) - cp[k * 6 + 3];
        const double ey = fma(R[1][0], sx, fma(R[1][1], sy, fma(R[1][2], sz, tv[1]))) - cp[k * 6 + 4];
        const double ez = fma(R[2][0], sx, fma(R[2][1], sy, fma(R[2][2], sz, tv[2]))) - cp[k * 6 + 5];
        const double d2 = fma(ex, ex, fma(ey, ey, ez * ez));
        if (d2 < A.thr2) { ++cnt; err += d2; }
    }
    int bc = valid ? cnt : -1, bi = valid ? it : 0x7fffffff;
    double be = valid ? err : 0.0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int oc = __shfl_xor(bc, m), oi = __shfl_xor(bi, m);
        const double oe = __shfl_xor(be, m);
        if (rs_better(oc, oe, oi, bc, be, bi)) { bc = oc; be = oe; bi = oi; }
    }
    if (lane == 0) { s_best[w].cnt = bc; s_best[w].err = be; s_best[w].it = bi; }
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < RS_BLOCK / 64; ++k)
            if (rs_better(s_best[k].cnt, s_best[k].err, s_best[k].it, bc, be, bi)) { bc = s_best[k].cnt; be = s_best[k].err; bi = s_best[k].it; }
        out->cnt = bc; out->err = be; out->it = bi;
    }
}

__global__ __launch_bounds__(64) void ransac_final_kernel(RsArgs A) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int K = min(A.count[pair], A.cap);
    int bc = -1, bi = 0x7fffffff;
    double be = 0.0;
    for (int k = lane; k < A.nblk; k += 64) {
        const RsBest b = A.blk[(size_t)pair * A.nblk + k];
        if (rs_better(b.cnt, b.err, b.it, bc, be, bi)) { bc = b.cnt; be = b.err; bi = b.it; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int oc = __shfl_xor(bc, m), oi = __shfl_xor(bi, m);
        const double oe = __shfl_xor(be, m);
        if (rs_better(oc, oe, oi, bc, be, bi)) { bc = oc; be = oe; bi = oi; }
    }
    if (lane != 0) return;
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, tv[3] = {0, 0, 0};
    const bool have = K >= 3 && bc > 0;      // a result without inliers never replaces Open3D's default (identity)
    if (have) {
        const uint64_t base = stream_base(A.seed, A.pair_ids ? (uint64_t)A.pair_ids[pair] : (uint64_t)pair);
        rs_fit(A.matches + (size_t)pair * A.cap * 3, K, base, bi, A.s_pcd + (size_t)pair * A.N * 3, A.t_pcd + (size_t)pair * A.M * 3,
               A.N, A.M, R, tv);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        A.trn[(size_t)pair * 3 + a] = tv[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) A.rot[(size_t)pair * 9 + a * 3 + b] = R[a][b];
    }
    if (A.fitness) A.fitness[pair] = have ? (double)bc / (double)K : 0.0;
    if (A.rmse) A.rmse[pair] = have ? sqrt(be / (double)bc) : 0.0;
    if (A.best_iter) A.best_iter[pair] = have ? bi : -1;
}

// ------------------------------------------------------------------------------------------------------------
// registration recall (Redwood benchmark error)
// ------------------------------------------------------------------------------------------------------------
__global__ void recall_kernel(int P, const double* __restrict__ Re, const double* __restrict__ te, const float* __restrict__ Rg,
                              const float* __restrict__ tg, const double* __restrict__ info, double thr2, double* __restrict__ err,
                              int* __restrict__ success) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double G[3][3], g[3], E[3][3], e[3];
    for (int a = 0; a < 3; ++a) {
        g[a] = (double)tg[p * 3 + a];
        e[a] = te[p * 3 + a];
        for (int b = 0; b < 3; ++b) { G[a][b] = (double)Rg[p * 9 + a * 3 + b]; E[a][b] = Re[p * 9 + a * 3 + b]; }
    }
    // inv([G g; 0 1]) = [G^-1, -G^-1 g]: a true inverse (adjugate), not the transpose -- the float32 ground truth is
    // orthonormal only to 1e-7 and np.linalg.inv (loss.py:438) inverts what it is given
    const double det = det3(G);
    double Gi[3][3];
    Gi[0][0] = (G[1][1] * G[2][2] - G[1][2] * G[2][1]) / det; Gi[0][1] = (G[0][2] * G[2][1] - G[0][1] * G[2][2]) / det;
    Gi[0][2] = (G[0][1] * G[1][2] - G[0][2] * G[1][1]) / det; Gi[1][0] = (G[1][2] * G[2][0] - G[1][0] * G[2][2]) / det;
    Gi[1][1] = (G[0][0] * G[2][2] - G[0][2] * G[2][0]) / det; Gi[1][2] = (G[0][2] * G[1][0] - G[0][0] * G[1][2]) / det;
    Gi[2][0] = (G[1][0] * G[2][1] - G[1][1] * G[2][0]) / det; Gi[2][1] = (G[0][1] * G[2][0] - G[0][0] * G[2][1]) / det;
    Gi[2][2] = (G[0][0] * G[1][1] - G[0][1] * G[1][0]) / det;
    double Q[3][3], tr[3];
    for (int a = 0; a < 3; ++a) {
        tr[a] = Gi[a][0] * (e[0] - g[0]) + Gi[a][1] * (e[1] - g[1]) + Gi[a][2] * (e[2] - g[2]);
        for (int b = 0; b < 3; ++b) Q[a][b] = Gi[a][0] * E[0][b] + Gi[a][1] * E[1][b] + Gi[a][2] * E[2][b];
    }
    // unit quaternion (w, x, y, z), w >= 0 -- nibabel.quaternions.mat2quat takes the dominant eigenvector of the
    // 4x4 K matrix; for a rotation matrix that is this closed form (largest-pivot branch for accuracy)
    double qw, qx, qy, qz;
    const double trc = Q[0][0] + Q[1][1] + Q[2][2];
    if (trc > 0.0) {
        const double s = sqrt(trc + 1.0) * 2.0;
        qw = 0.25 * s; qx = (Q[2][1] - Q[1][2]) / s; qy = (Q[0][2] - Q[2][0]) / s; qz = (Q[1][0] - Q[0][1]) / s;
    } else if (Q[0][0] > Q[1][1] && Q[0][0] > Q[2][2]) {
        const double s = sqrt(1.0 + Q[0][0] - Q[1][1] - Q[2][2]) * 2.0;
        qw = (Q[2][1] - Q[1][2]) / s; qx = 0.25 * s; qy = (Q[0][1] + Q[1][0]) / s; qz = (Q[0][2] + Q[2][0]) / s;
    } else if (Q[1][1] > Q[2][2]) {
        const double s = sqrt(1.0 + Q[1][1] - Q[0][0] - Q[2][2]) * 2.0;
        qw = (Q[0][2] - Q[2][0]) / s; qx = (Q[0][1] + Q[1][0]) / s; qy = 0.25 * s; qz = (Q[1][2] + Q[2][1]) / s;
    } else {
        const double s = sqrt(1.0 + Q[2][2] - Q[0][0] - Q[1][1]) * 2.0;
        qw = (Q[1][0] - Q[0][1]) / s; qx = (Q[0][2] + Q[2][0]) / s; qy = (Q[1][2] + Q[2][1]) / s; qz = 0.25 * s;
    }
    const double qn = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    const double sg = qw < 0.0 ? -qn : qn;
    const double er[6] = {tr[0], tr[1], tr[2], qx * sg, qy * sg, qz * sg};
    const double* I = info + (size_t)p * 36;
    double v = 0.0;
    for (int a = 0; a < 6; ++a) {
        double r = 0.0;
        for (int b = 0; b < 6; ++b) r += I[a * 6 + b] * er[b];
        v += er[a] * r;
    }
    v /= I[0];
    err[p] = v;
    success[p] = v <= thr2 ? 1 : 0;
}

}  // namespace dr

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
using namespace dr;

extern "C" {

int dr_inlier_ratio_f32(int P, int cap, int N, int M, const int64_t* matches, const int32_t* count, const float* s_pcd,
                        const float* t_pcd, const float* rot, const float* trn, const float* s2t_flow, float inlier_thr,
                        float* ir, int32_t* n_inlier, void* stream) {
    if (P < 0 || cap < 0 || N <= 0 || M <= 0) return DR_EINVAL;
    if (P == 0) return DR_OK;
    if (!matches || !count || !s_pcd || !t_pcd || !rot || !trn || !ir || !n_inlier) return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DR_HIP_CHECK(hipMemsetAsync(n_inlier, 0, sizeof(int32_t) * P, st));
    // `inlier_thr ** 2` is a Python double; compared with a float32 tensor it is rounded to float32 (loss.py:397)
    const float thr2 = (float)((double)inlier_thr * (double)inlier_thr);
    if (cap > 0) {
        hipLaunchKernelGGL(inlier_kernel, dim3((cap + 255) / 256, P), dim3(256), 0, st, (const long long*)matches, count, cap, N, M,
                           s_pcd, t_pcd, rot, trn, s2t_flow, thr2, n_inlier);
        DR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ratio_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, (const int*)n_inlier, count, (const int*)nullptr, 3,
                       ir);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_nrfmr_f32(int P, int cap, int N, int M, const int64_t* matches, const int32_t* count, const float* s_pcd,
                 const float* t_pcd, const float* raw_pcd, const float* raw_flow, const int32_t* raw_offsets,
                 const int64_t* metric_index, const int32_t* q_offsets, int max_q, const float* rot, const float* trn,
                 float knn_radius, float recall_thr, float* nrfmr, int32_t* n_recalled, float* blended, void* stream) {
    if (P < 0 || cap < 0 || N <= 0 || M <= 0 || max_q < 0) return DR_EINVAL;
    if (P == 0) return DR_OK;
    if (!matches || !count || !s_pcd || !t_pcd || !raw_pcd || !raw_flow || !raw_offsets || !metric_index || !q_offsets || !rot ||
        !trn || !nrfmr || !n_recalled)
        return DR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DR_HIP_CHECK(hipMemsetAsync(n_recalled, 0, sizeof(int32_t) * P, st));
    if (max_q > 0 && cap > 0) {
        NrArgs A{(const long long*)matches, count, cap, N, M, s_pcd, t_pcd, raw_pcd, raw_flow, raw_offsets,
                 (const long long*)metric_index, q_offsets, rot, trn, knn_radius, recall_thr, n_recalled, blended};
        hipLaunchKernelGGL(nrfmr_kernel, dim3((max_q + 255) / 256, P), dim3(256), 0, st, A);
        DR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ratio_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, (const int*)n_recalled, (const int*)nullptr,
                       q_offsets, 1, nrfmr);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

size_t dr_ransac_workspace_bytes(int P, int cap, int iters) {
    if (P <= 0 || iters <= 0 || cap <= 0) return 0;
    return (size_t)P * ((iters + RS_BLOCK - 1) / RS_BLOCK) * sizeof(RsBest) + (size_t)P * cap * 6 * sizeof(double);
}

int dr_ransac_corr_f64(int P, int cap, int N, int M, const int64_t* matches, const int32_t* count, const float* s_pcd,
                       const float* t_pcd, double distance_thr, int iters, uint64_t seed, const int64_t* pair_ids, double* rot,
                       double* trn, double* fitness, double* inlier_rmse, int32_t* best_iter, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (P < 0 || cap <= 0 || N <= 0 || M <= 0 || iters <= 0 || !(distance_thr > 0.0)) return DR_EINVAL;
    if (P == 0) return DR_OK;
    if (!matches || !count || !s_pcd || !t_pcd || !rot || !trn) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_ransac_workspace_bytes(P, cap, iters)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    RsArgs A{};
    A.matches = (const long long*)matches; A.count = count; A.cap = cap; A.N = N; A.M = M;
    A.s_pcd = s_pcd; A.t_pcd = t_pcd; A.pair_ids = (const long long*)pair_ids;
    A.thr2 = distance_thr * distance_thr; A.iters = iters; A.seed = seed;
    A.blk = (RsBest*)workspace; A.nblk = (iters + RS_BLOCK - 1) / RS_BLOCK;
    A.pts = (double*)((char*)workspace + (size_t)P * A.nblk * sizeof(RsBest));
    A.rot = rot; A.trn = trn; A.fitness = fitness; A.rmse = inlier_rmse; A.best_iter = best_iter;
    hipLaunchKernelGGL(ransac_gather_kernel, dim3((cap + 255) / 256, P), dim3(256), 0, st, A);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ransac_eval_kernel, dim3(A.nblk, P), dim3(RS_BLOCK), 0, st, A);
    DR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ransac_final_kernel, dim3(P), dim3(64), 0, st, A);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int dr_registration_recall_f64(int P, const double* rot_est, const double* trn_est, const float* rot_gt, const float* trn_gt,
                               const double* info, double thr, double* err, int32_t* success, void* stream) {
    if (P < 0) return DR_EINVAL;
    if (P == 0) return DR_OK;
    if (!rot_est || !trn_est || !rot_gt || !trn_gt || !info || !err || !success) return DR_EINVAL;
    hipLaunchKernelGGL(recall_kernel, dim3((P + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, rot_est, trn_est, rot_gt, trn_gt, info,
                       thr * thr, err, (int*)success);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // extern "C"
