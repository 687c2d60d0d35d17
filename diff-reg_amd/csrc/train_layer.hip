// train_layer.hip -- GeometryAttentionLayer (3D/models/transformero.py:43-96) for TRAINING as two entry points: the forward that keeps what its
// backward needs, and the whole backward -- every kernel of it launched from here.  Round 4 drove the same kernels one by one from Python
// (diffreg_hip/autograd.py: ~20 calls forward, ~55 backward per layer call, 20 layer calls per step); a training step was ~12 ms of kernels inside
// 28 ms of host time.  Nothing new is computed here: the projections and their gradients run on the library's f32-input MFMA GEMM (launch_gemm),
// the attention on dr_attention_f32 / dr_attention_backward_f32 (flash-style, no [B,H,L,S] matrix), LayerNorm / ReLU / rotary on the kernels of
// train.hip.  New kernels: a batched transposition (the GEMM contracts along contiguous k: x W for a gradient w.r.t. the input and g^T x for a weight
// gradient need the transposed operand) and a two-operand add.
#include <string.h>
#include "kernels.h"

namespace dr {
namespace {

// out[c][r] = src[r][c] for up to 12 matrices in one launch; a destination row has stride ld_dst; its first w_dst >= rows entries are written,
// zero behind `rows` (the GEMM wants its k extent -- here the token count -- a multiple of 4)
struct TrProblem { const float* src; float* dst; int rows, cols, ld_src, ld_dst, w_dst, tile0, tiles_c; };
struct TrBatch { TrProblem p[12]; int n; };
__global__ __launch_bounds__(256) void transpose_batch_kernel(TrBatch G) {
    __shared__ float tile[32][33];
    int pi = 0;
    while (pi + 1 < G.n && (int)blockIdx.x >= G.p[pi + 1].tile0) ++pi;
    const TrProblem& P = G.p[pi];
    const int tl = blockIdx.x - P.tile0, tr = tl / P.tiles_c, tc = tl % P.tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
        tile[ty + 8 * k][tx] = (r < P.rows && c < P.cols) ? P.src[(size_t)r * P.ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tc * 32 + ty + 8 * k, r = tr * 32 + tx;
        if (c < P.cols && r < P.w_dst) P.dst[(size_t)c * P.ld_dst + r] = tile[tx][ty + 8 * k];
    }
}
struct Transposer {
    TrBatch g;
    int tiles;
    bool overflow;                                                        // an add() beyond the batch's 12 slots: launch() then fails instead of overrunning the struct
    static constexpr int CAP = (int)(sizeof(TrBatch::p) / sizeof(TrProblem));
    Transposer() { memset(&g, 0, sizeof(g)); tiles = 0; overflow = false; }
    void add(const float* src, int rows, int cols, int ld_src, float* dst, int ld_dst, int w_dst = -1) {
        if (g.n >= CAP) { overflow = true; return; }
        TrProblem& p = g.p[g.n++];
        p.src = src; p.dst = dst; p.rows = rows; p.cols = cols; p.ld_src = ld_src; p.ld_dst = ld_dst; p.w_dst = w_dst < 0 ? ld_dst : w_dst;
        p.tile0 = tiles; p.tiles_c = (cols + 31) / 32;
        tiles += ((p.w_dst + 31) / 32) * p.tiles_c;                       // (the pad entries of a destination row are covered too)
    }
    int launch(hipStream_t st) {
        if (overflow) return DR_EINVAL;
        if (g.n == 0) return DR_OK;
        hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles), dim3(256), 0, st, g);
        DR_LAUNCH_CHECK();
        memset(&g, 0, sizeof(g)); tiles = 0;
        return DR_OK;
    }
};

__global__ __launch_bounds__(256) void add2_kernel(long long n4, const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    const float4 x = a[e], y = b[e];
    out[e] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

struct Carve {
    char* base; size_t off;
    explicit Carve(void* p) : base((char*)p), off(0) {}
    float* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += n * sizeof(float);
        return r;
    }
};
inline int up4(int x) { return (x + 3) & ~3; }

// what the forward keeps (floats): q, k (rotary applied), v, the heads' output o, merge(o) before norm1, norm1's output m, the hidden
// activation h, mlp.2(h) before norm2, and the two LayerNorms' (mean, rstd) rows
struct Saved {
    float *qw, *kw, *vw, *o, *m_pre, *m, *h, *f_pre, *st1, *st2;
    static size_t carve(void* buf, Saved& s, size_t R, size_t Q, int C) {
        Carve c(buf);
        s.qw = c.take(R * C); s.kw = c.take(Q * C); s.vw = c.take(Q * C); s.o = c.take(R * C); s.m_pre = c.take(R * C); s.m = c.take(R * C);
        s.h = c.take(R * 2 * C); s.f_pre = c.take(R * C); s.st1 = c.take(2 * R); s.st2 = c.take(2 * R);
        return c.off + 256;
    }
};

inline void gemm_problem(GemmProblem& p, const float* A, int lda, const float* A2, int lda2, int K1, const float* W, float* out, int ldo, int rows,
                         int ncols, int K, int epi, const float* addend) {
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.A2 = A2; p.lda2 = lda2; p.K1 = K1; p.W = W; p.out = out; p.ldo = ldo; p.rows = rows; p.ncols = ncols; p.K = K;
    p.epi = epi; p.scale = 1.f; p.addend = addend;
}

}  // namespace
}  // namespace dr

using namespace dr;

extern "C" {

size_t dr_attention_layer_train_saved_bytes(int B, int L, int S, int C) {
    if (B < 1 || L < 1 || S < 1 || C < 4) return 0;
    Saved s;
    return Saved::carve(nullptr, s, (size_t)B * L, (size_t)B * S, C);
}

int dr_attention_layer_train_forward_f32(const dr_layer_weights* w, int C, int H, int B, int L, int S, const float* x, const float* y,
                                         const float* cos_x, const float* sin_x, const float* cos_y, const float* sin_y,
                                         const uint8_t* x_mask, const uint8_t* y_mask, float* out, void* saved, size_t saved_bytes, void* stream) {
    if (!w || !x || !y || !cos_x || !sin_x || !cos_y || !sin_y || !out || !saved || B < 1 || L < 1 || S < 1 || C % H || (C / H) % 4 || C % 4) return DR_EINVAL;
    if ((x_mask == nullptr) != (y_mask == nullptr)) return DR_EINVAL;
    if (saved_bytes < dr_attention_layer_train_saved_bytes(B, L, S, C)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int R = B * L, Q = B * S, d = C / H;
    Saved sv;
    Saved::carve(saved, sv, R, Q, C);
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    // q | k | v: one launch; the rotary code in the GEMM's epilogue (transformero.py:61-70)
    gemm_problem(g.p[0], x, C, nullptr, 0, C, w->q_proj, sv.qw, C, R, C, C, EPI_ROTARY, nullptr);
    g.p[0].cosT = cos_x; g.p[0].sinT = sin_x; g.p[0].rot_C = C;
    gemm_problem(g.p[1], y, C, nullptr, 0, C, w->k_proj, sv.kw, C, Q, C, C, EPI_ROTARY, nullptr);
    g.p[1].cosT = cos_y; g.p[1].sinT = sin_y; g.p[1].rot_C = C;
    gemm_problem(g.p[2], y, C, nullptr, 0, C, w->v_proj, sv.vw, C, Q, C, C, EPI_NONE, nullptr);
    g.n = 3;
    int rc = launch_gemm(g, st);
    if (rc) return rc;
    rc = dr_attention_f32(B, H, L, S, d, sv.qw, sv.kw, sv.vw, C, x_mask, y_mask, 1.0f / sqrtf((float)d), sv.o, stream);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], sv.o, C, nullptr, 0, C, w->merge, sv.m_pre, C, R, C, C, EPI_NONE, nullptr);
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    rc = dr_layernorm_f32(R, C, sv.m_pre, w->norm1_w, w->norm1_b, 1e-5f, sv.m, sv.st1, stream);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], x, C, sv.m, C, C, w->mlp0, sv.h, 2 * C, R, 2 * C, 2 * C, EPI_RELU, nullptr);      // mlp.0(cat[x, message]) + ReLU
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], sv.h, 2 * C, nullptr, 0, 2 * C, w->mlp2, sv.f_pre, C, R, C, 2 * C, EPI_NONE, nullptr);
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    // out = x + norm2(.)  (norm2's output is not needed again: it lands in `out` and the residual is added in place)
    rc = dr_layernorm_f32(R, C, sv.f_pre, w->norm2_w, w->norm2_b, 1e-5f, out, sv.st2, stream);
    if (rc) return rc;
    const long long n4 = (long long)R * C / 4;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, n4, (const float4*)out, (const float4*)x, (float4*)out);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

size_t dr_attention_layer_backward_workspace_bytes(int B, int H, int L, int S, int C) {
    if (B < 1 || L < 1 || S < 1 || C < 4 || H < 1) return 0;
    const size_t R = (size_t)B * L, Q = (size_t)B * S, R4 = up4((int)R), Q4 = up4((int)Q);
    Carve c(nullptr);
    c.take(R * C); c.take(R * 2 * C); c.take(R * C); c.take(R * C); c.take(R * C); c.take(R * C);   // g_fpre, g_h, g_x1, g_m, g_mpre, g_o
    c.take(R * C); c.take(Q * C); c.take(Q * C); c.take(R * C); c.take(Q * C);                      // g_qw, g_kw, g_vw, g_qpre, g_kpre
    c.take((size_t)2 * C * C); c.take((size_t)4 * C * C); c.take((size_t)C * C); c.take((size_t)C * C); c.take((size_t)2 * C * C);   // W2^T, W0^T, Wm^T, Wq^T, [Wk^T | Wv^T]
    c.take(C * R4); c.take(2 * C * R4); c.take(2 * C * R4); c.take(2 * C * R4); c.take(C * R4); c.take(C * R4); c.take(C * R4);   // transposed activations
    c.take(C * Q4); c.take(C * Q4); c.take(C * Q4);
    c.take(dr_layernorm_backward_workspace_bytes(C) / sizeof(float));
    c.take(dr_attention_backward_workspace_bytes(B, H, L) / sizeof(float) + 64);
    return c.off + 256;
}

int dr_attention_layer_backward_f32(const dr_layer_weights* w, int C, int H, int B, int L, int S, const float* x, const float* y,
                                    const float* cos_x, const float* sin_x, const float* cos_y, const float* sin_y, const uint8_t* x_mask,
                                    const uint8_t* y_mask, const void* saved, const float* grad_out, float* grad_x, float* grad_y,
                                    const dr_layer_grads* gw, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || !gw || !x || !y || !cos_x || !sin_x || !cos_y || !sin_y || !saved || !grad_out || !grad_x || !grad_y || B < 1 || L < 1 || S < 1 ||
        C % H || (C / H) % 4 || C % 4)
        return DR_EINVAL;
    if ((x_mask == nullptr) != (y_mask == nullptr)) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_attention_layer_backward_workspace_bytes(B, H, L, S, C)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int R = B * L, Q = B * S, d = C / H, R4 = up4(R), Q4 = up4(Q), C2 = 2 * C;
    Saved sv;
    Saved::carve(const_cast<void*>(saved), sv, R, Q, C);
    Carve c(workspace);
    float *g_fpre = c.take((size_t)R * C), *g_h = c.take((size_t)R * C2), *g_x1 = c.take((size_t)R * C), *g_m = c.take((size_t)R * C),
          *g_mpre = c.take((size_t)R * C), *g_o = c.take((size_t)R * C);
    float *g_qw = c.take((size_t)R * C), *g_kw = c.take((size_t)Q * C), *g_vw = c.take((size_t)Q * C), *g_qpre = c.take((size_t)R * C),
          *g_kpre = c.take((size_t)Q * C);
    float *TW2 = c.take((size_t)C2 * C), *TW0 = c.take((size_t)C2 * C2), *TWm = c.take((size_t)C * C), *TWq = c.take((size_t)C * C),
          *TWkv = c.take((size_t)C * C2);
    float *T_gf = c.take((size_t)C * R4), *T_h = c.take((size_t)C2 * R4), *T_gh = c.take((size_t)C2 * R4), *T_cat = c.take((size_t)C2 * R4),
          *T_gm = c.take((size_t)C * R4), *T_o = c.take((size_t)C * R4), *T_gq = c.take((size_t)C * R4);
    float *T_gk = c.take((size_t)C * Q4), *T_gv = c.take((size_t)C * Q4), *T_y = c.take((size_t)C * Q4);
    float* ln_ws = c.take(dr_layernorm_backward_workspace_bytes(C) / sizeof(float));
    const size_t att_wsb = dr_attention_backward_workspace_bytes(B, H, L);
    float* att_ws = c.take(att_wsb / sizeof(float) + 64);
    int rc;
    // ---- everything that only needs the forward's tensors is transposed first, in one launch
    Transposer T;
    T.add(w->mlp2, C, C2, C2, TW2, C);                 // W2 [C, 2C]  -> [2C, C]
    T.add(w->mlp0, C2, C2, C2, TW0, C2);               // W0 [2C, 2C] -> its transpose
    T.add(w->merge, C, C, C, TWm, C);
    T.add(w->q_proj, C, C, C, TWq, C);
    T.add(w->k_proj, C, C, C, TWkv, C2, C);            // [Wk^T | Wv^T]: row j = (Wk[:, j], Wv[:, j])
    T.add(w->v_proj, C, C, C, TWkv + C, C2, C);
    T.add(sv.h, R, C2, C2, T_h, R4);
    T.add(x, R, C, C, T_cat, R4);                      // cat[x, m]^T = [x^T ; m^T]
    T.add(sv.m, R, C, C, T_cat + (size_t)C * R4, R4);
    T.add(sv.o, R, C, C, T_o, R4);
    // self-attention calls (y is x): y^T is x^T, already the first C rows of cat^T
    const bool y_is_x = y == x && Q == R;
    if (y_is_x) T_y = T_cat;
    else T.add(y, Q, C, C, T_y, Q4);
    rc = T.launch(st);
    if (rc) return rc;
    // ---- norm2 -> mlp.2 -> ReLU -> mlp.0
    rc = dr_layernorm_backward_f32(R, C, sv.f_pre, w->norm2_w, sv.st2, grad_out, g_fpre, gw->norm2_w, gw->norm2_b, ln_ws, stream);
    if (rc) return rc;
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], g_fpre, C, nullptr, 0, C, TW2, g_h, C2, R, C2, C, EPI_NONE, nullptr);                 // g W2
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    rc = dr_relu_backward_f32((long long)R * C2, sv.h, g_h, g_h, stream);
    if (rc) return rc;
    T.add(g_fpre, R, C, C, T_gf, R4);
    T.add(g_h, R, C2, C2, T_gh, R4);
    rc = T.launch(st);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], g_h, C2, nullptr, 0, C2, TW0, g_x1, C, R, C, C2, EPI_NONE, grad_out);                  // grad_out + (g_h W0)[:, :C]   (the residual + cat's x half)
    gemm_problem(g.p[1], g_h, C2, nullptr, 0, C2, TW0 + (size_t)C * C2, g_m, C, R, C, C2, EPI_NONE, nullptr);  // (g_h W0)[:, C:]  -> norm1's output
    gemm_problem(g.p[2], T_gf, R4, nullptr, 0, R4, T_h, gw->mlp2, C2, C, C2, R4, EPI_NONE, nullptr);           // g^T h
    gemm_problem(g.p[3], T_gh, R4, nullptr, 0, R4, T_cat, gw->mlp0, C2, C2, C2, R4, EPI_NONE, nullptr);        // g_h^T cat[x, m]
    g.n = 4;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    // ---- norm1 -> merge
    rc = dr_layernorm_backward_f32(R, C, sv.m_pre, w->norm1_w, sv.st1, g_m, g_mpre, gw->norm1_w, gw->norm1_b, ln_ws, stream);
    if (rc) return rc;
    T.add(g_mpre, R, C, C, T_gm, R4);
    rc = T.launch(st);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], g_mpre, C, nullptr, 0, C, TWm, g_o, C, R, C, C, EPI_NONE, nullptr);
    gemm_problem(g.p[1], T_gm, R4, nullptr, 0, R4, T_o, gw->merge, C, C, C, R4, EPI_NONE, nullptr);
    g.n = 2;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    // ---- attention, rotary code, projections
    rc = dr_attention_backward_f32(B, H, L, S, d, sv.qw, sv.kw, sv.vw, sv.o, g_o, C, x_mask, y_mask, 1.0f / sqrtf((float)d), g_qw, g_kw, g_vw, att_ws,
                                   att_wsb, stream);
    if (rc) return rc;
    rc = dr_rotary_f32(R, C, g_qw, cos_x, sin_x, 1, 1.f, g_qpre, stream);
    if (rc) return rc;
    rc = dr_rotary_f32(Q, C, g_kw, cos_y, sin_y, 1, 1.f, g_kpre, stream);
    if (rc) return rc;
    T.add(g_qpre, R, C, C, T_gq, R4);
    T.add(g_kpre, Q, C, C, T_gk, Q4);
    T.add(g_vw, Q, C, C, T_gv, Q4);
    rc = T.launch(st);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], g_qpre, C, nullptr, 0, C, TWq, grad_x, C, R, C, C, EPI_NONE, g_x1);                    // + the residual / mlp part
    gemm_problem(g.p[1], g_kpre, C, g_vw, C, C, TWkv, grad_y, C, Q, C, C2, EPI_NONE, nullptr);                  // g_k Wk + g_v Wv
    g.n = 2;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    gemm_problem(g.p[0], T_gq, R4, nullptr, 0, R4, T_cat, gw->q_proj, C, C, C, R4, EPI_NONE, nullptr);          // g_q^T x   (x^T = the first C rows of cat^T)
    gemm_problem(g.p[1], T_gk, Q4, nullptr, 0, Q4, T_y, gw->k_proj, C, C, C, Q4, EPI_NONE, nullptr);
    gemm_problem(g.p[2], T_gv, Q4, nullptr, 0, Q4, T_y, gw->v_proj, C, C, C, Q4, EPI_NONE, nullptr);
    g.n = 3;
    return launch_gemm(g, st);
}

}  // extern "C"
