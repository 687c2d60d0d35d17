// procrustes.hip -- SoftProcrustesLayer on device: top-K of the N*M confidences, weighted Kabsch,
// 3x3 SVD in fp64, condition-number gate.  Replaces 3D/models/procrustes.py:17-93 including the
// per-step `.cpu().double().svd()` round trip (procrustes.py:35-36, quirk Q13): no host sync.
//
// One workgroup per pair:
//   1. radix select (4 x 8-bit histogram passes over the tile, L2 resident) -> key of the K-th largest
//   2. deterministic compaction in index order (ballot ranks) of the entries > tau plus the first
//      entries == tau  (torch.sort is unstable, ties are unspecified upstream: lowest index first)
//   3. fp64 sums  W1 = sum|w|, sum w X, sum w Y, sum w Y X^T  ->  Sxy = sum w^ Y X^T - (2 - s) Ybar Xbar^T
//      with w^ = w / (W1 + eps), s = sum w^  (identical to (Y - Ybar)^T (w^ (X - Xbar)), procrustes.py:27-33)
//   4. one-sided Jacobi SVD, R = U diag(1,1,det U det V) V^T, t = Ybar - R Xbar, cond = Dmax / Dmin
#include "kernels.h"

namespace dr {

constexpr int PK_MAX = 4096;   // largest K (= max(N, M) * sample_rate) supported

__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // larger float <=> larger key
}

__device__ void svd3_jacobi(double A[3][3], double U[3][3], double S[3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-300) continue;
                off = fmax(off, fabs(ga) / sqrt(al * be + 1e-300));
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < 3; ++j) S[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    // sort descending (columns of A and V move together)
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (S[b] > S[a]) {
                double ts = S[a]; S[a] = S[b]; S[b] = ts;
                for (int i = 0; i < 3; ++i) {
                    double ta = A[i][a]; A[i][a] = A[i][b]; A[i][b] = ta;
                    double tv = V[i][a]; V[i][a] = V[i][b]; V[i][b] = tv;
                }
            }
    const double tiny = 1e-200;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 3; ++i) U[i][j] = S[j] > tiny ? A[i][j] / S[j] : (i == j ? 1.0 : 0.0);
    if (S[2] > 1e-14 * S[0] && S[2] > tiny) {
        for (int i = 0; i < 3; ++i) U[i][2] = A[i][2] / S[2];
    } else {   // rank deficient: complete the basis (cond = inf/huge rejects it unless the gate is open)
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}

__device__ __forceinline__ double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
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
    int* topk_idx;            // optional [P, K] flat indices of the selected entries (index order)
    int N, M, K_fixed, use_mask_len;
    float sample_rate, max_cond;
};

__global__ __launch_bounds__(1024) void procrustes_kernel(ProcArgs A) {
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix, s_remaining;
    __shared__ int s_wcnt[2][16];
    __shared__ int s_sel[PK_MAX];
    __shared__ double s_red[16][16];
    __shared__ int s_len[2];

    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int N = A.N, M = A.M, NM = N * M;
    const float* conf = A.conf + (size_t)pair * NM;

    // ---- K (procrustes.py:61-65; 4D/models/procrustes.py:61-62 uses the mask sums, quirk Q17) -------
    int K = A.K_fixed;
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

    // ---- radix select of the K-th largest key ----------------------------------------------------------
    unsigned prefix = 0, mask = 0, remaining = (unsigned)K;
    if (K > 0) {
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (t < 256) s_hist[t] = 0;
            __syncthreads();
            for (int e = t; e < NM; e += 1024) {
                const unsigned key = order_key(conf[e]);
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (t == 0) {
                unsigned cum = 0;
                int b = 255;
                for (; b > 0; --b) {
                    if (cum + s_hist[b] >= remaining) break;
                    cum += s_hist[b];
                }
                s_prefix = prefix | ((unsigned)b << shift);
                s_remaining = remaining - cum;
            }
            __syncthreads();
            prefix = s_prefix;
            remaining = s_remaining;
            mask |= 0xFFu << shift;
            __syncthreads();
        }
    }
    const unsigned tau = prefix;            // key of the K-th largest entry; take `remaining` entries == tau
    const int n_gt_total = K - (int)remaining;

    // ---- compaction in index order ------------------------------------------------------------------------
    const int per_wave = ((NM + 15) / 16 + 63) / 64 * 64;
    const int beg = w * per_wave, end = (beg + per_wave < NM) ? beg + per_wave : NM;
    int cg = 0, ce = 0;
    for (int base = beg; base < end; base += 64) {
        const int e = base + lane;
        const unsigned key = e < end ? order_key(conf[e]) : 0u;
        const bool gt = e < end && key > tau, eq = e < end && key == tau;
        cg += __popcll(__ballot(gt));
        ce += __popcll(__ballot(eq));
    }
    if (lane == 0) { s_wcnt[0][w] = cg; s_wcnt[1][w] = ce; }
    __syncthreads();
    int off_g = 0, off_e = 0;
    for (int k = 0; k < w; ++k) { off_g += s_wcnt[0][k]; off_e += s_wcnt[1][k]; }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int base = beg; base < end && K > 0; base += 64) {
        const int e = base + lane;
        const unsigned key = e < end ? order_key(conf[e]) : 0u;
        const bool gt = e < end && key > tau, eq = e < end && key == tau;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        if (gt) s_sel[off_g + __popcll(bg & lt_mask)] = e;
        if (eq) {
            const int rank = off_e + __popcll(be & lt_mask);
            if (rank < (int)remaining) s_sel[n_gt_total + rank] = e;
        }
        off_g += __popcll(bg);
        off_e += __popcll(be);
    }
    __syncthreads();

    // ---- weighted sums in fp64 ---------------------------------------------------------------------------
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    const float* Xs = A.src_pcd + (size_t)pair * N * 3;
    const float* Ys = A.tgt_pcd + (size_t)pair * M * 3;
    for (int s = t; s < K; s += 1024) {
        const int e = s_sel[s];
        const double wv = (double)conf[e];
        const int i = e / M, j = e % M;
        const double x0 = Xs[i * 3], x1 = Xs[i * 3 + 1], x2 = Xs[i * 3 + 2];
        const double y0 = Ys[j * 3], y1 = Ys[j * 3 + 1], y2 = Ys[j * 3 + 2];
        acc[0] += fabs(wv);
        acc[1] += wv * x0; acc[2] += wv * x1; acc[3] += wv * x2;
        acc[4] += wv * y0; acc[5] += wv * y1; acc[6] += wv * y2;
        acc[7] += wv * y0 * x0; acc[8] += wv * y0 * x1; acc[9] += wv * y0 * x2;
        acc[10] += wv * y1 * x0; acc[11] += wv * y1 * x1; acc[12] += wv * y1 * x2;
        acc[13] += wv * y2 * x0; acc[14] += wv * y2 * x1; acc[15] += wv * y2 * x2;
        if (A.topk_idx) A.topk_idx[(size_t)pair * K + s] = e;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double v = wave_sum(acc[i]);
        if (lane == 0) s_red[w][i] = v;
    }
    __syncthreads();
    if (t != 0) return;
    double sum[16];
    for (int i = 0; i < 16; ++i) {
        double v = 0;
        for (int k = 0; k < 16; ++k) v += s_red[k][i];
        sum[i] = v;
    }
    const double inv = 1.0 / (sum[0] + 1e-4);        // eps of batch_weighted_procrustes
    const double sw = sum[0] * inv;                   // sum of normalised weights (w >= 0 here)
    double mx[3] = {sum[1] * inv, sum[2] * inv, sum[3] * inv};
    double my[3] = {sum[4] * inv, sum[5] * inv, sum[6] * inv};
    double Sxy[3][3], U[3][3], V[3][3], D[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Sxy[a][b] = sum[7 + 3 * a + b] * inv - (2.0 - sw) * my[a] * mx[b];
    // the reference forms Sxy in fp32 before `.double()` (procrustes.py:33-34)
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Sxy[a][b] = (double)(float)Sxy[a][b];
    double Aw[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Aw[a][b] = Sxy[a][b];
    svd3_jacobi(Aw, U, D, V);
    const double cond = D[0] / D[2];
    const double dd = det3(U) * det3(V);
    double Rm[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rm[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + dd * U[a][2] * V[b][2];
    float Rf32[9], tf32[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rf32[a * 3 + b] = (float)Rm[a][b];
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
    for (int i = 0; i < 9; ++i) {
        R[i] = Rf32[i];
        Rf[i] = good ? Rf32[i] : ((i % 4 == 0) ? 1.f : 0.f);
    }
    for (int i = 0; i < 3; ++i) {
        tt[i] = tf32[i];
        tf[i] = good ? tf32[i] : 0.f;
    }
    A.cond[pair] = cond;
    A.ok[pair] = good ? 1 : 0;
}

int launch_procrustes(const float* conf, const float* src_pcd, const float* tgt_pcd, const uint8_t* src_mask,
                      const uint8_t* tgt_mask, int P, int N, int M, int use_mask_len, float sample_rate, float max_cond,
                      float* R, float* t, float* Rf, float* tf, double* cond, int* ok, int* topk_idx, hipStream_t st) {
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
    ProfScope ps(PK_PROCRUSTES, (double)P * N * M * 4.0, st);
    hipLaunchKernelGGL(procrustes_kernel, dim3(P), dim3(1024), 0, st, a);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

}  // namespace dr
