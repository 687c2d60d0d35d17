// loop2d3d.hip -- orchestration of the 2D-3D variant: CrossModalFusionModule denoiser,
// position-free matching head, reverse sampling of MATR2D3D.forward.  Same kernels as loop.hip
// (GEMM with bias / addend epilogues, attention with d_head = 64, post-LN LayerNorm, Sinkhorn,
// Procrustes, DDIM), enqueued on one stream without host synchronisation.
//
// Token layout: image patches of all P pairs (pair p at row p*M), then point nodes (P*M + p*N).
#include "kernels.h"
#include <string.h>

namespace dr {

struct Carver2 {
    char* base; size_t off;
    explicit Carver2(void* p) : base((char*)p), off(0) {}
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
};

struct F2Ws {
    float *tok0, *ta, *tb, *qkv, *att, *lin, *z, *hid, *f, *feat, *proj, *sim, *x0, *wconf, *cat, *emb2, *emb3, *warped;
    float *R, *t, *Rf, *tf;
    double *x, *cond;
    int* ok;
    void* skws; size_t skws_bytes;
    void* pws; size_t pws_bytes;
    static size_t carve(Carver2& c, F2Ws& w, const dr_loop2d3d_config& cfg, int P, int N, int M) {
        const size_t T = (size_t)P * (N + M), C = cfg.C, NM = (size_t)P * N * M, PM = (size_t)P * M, PN = (size_t)P * N;
        w.tok0 = c.take<float>(T * C); w.ta = c.take<float>(T * C); w.tb = c.take<float>(T * C);
        w.qkv = c.take<float>(T * 3 * C); w.att = c.take<float>(T * C); w.lin = c.take<float>(T * C);
        w.z = c.take<float>(T * C); w.hid = c.take<float>(T * 2 * C); w.f = c.take<float>(T * C);
        w.feat = c.take<float>(T * C); w.proj = c.take<float>(T * C);
        w.sim = c.take<float>(NM); w.x0 = c.take<float>(NM); w.wconf = c.take<float>(NM);
        w.cat = c.take<float>(PM * 2 * C); w.emb2 = c.take<float>(PM * 44); w.emb3 = c.take<float>(PN * 64);
        w.warped = c.take<float>(PN * 3);
        w.R = c.take<float>((size_t)P * 9); w.t = c.take<float>((size_t)P * 3);
        w.Rf = c.take<float>((size_t)P * 9); w.tf = c.take<float>((size_t)P * 3);
        w.x = c.take<double>(NM); w.cond = c.take<double>(P); w.ok = c.take<int>(P);
        const int strict = (cfg.flags & DR_LOOP_STRICT_F64) ? DR_SK_STRICT : 0;
        size_t a = dr_sinkhorn_workspace_bytes(P, N, M, 8, strict), b = dr_sinkhorn_workspace_bytes(P, N, M, 4, 0);
        w.skws_bytes = a > b ? a : b;
        w.skws = w.skws_bytes ? (void*)c.take<char>(w.skws_bytes) : nullptr;
        w.pws_bytes = procrustes_workspace_bytes(P, N, M);
        w.pws = w.pws_bytes ? (void*)c.take<char>(w.pws_bytes) : nullptr;
        return c.off + 256;
    }
};

static int gemm1(const float* A, int lda, const float* W, const float* bias, float* out, int ldo, int rows, int ncols, int K,
                 int epi, float scale, const float* addend, hipStream_t st) {
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    GemmProblem& p = g.p[0];
    p.A = A; p.W = W; p.out = out; p.rows = rows; p.ncols = ncols; p.K = K; p.K1 = K; p.lda = lda; p.ldo = ldo;
    p.epi = epi; p.scale = scale; p.bias = bias; p.addend = addend;
    g.n = 1;
    return launch_gemm(g, st);
}

// one vision3d TransformerLayer: x rows [xr0, +xrows) of xin attend y rows [yr0, +yrows) of yin
static int fusion_layer(const dr_fusion_layer_weights& W, int C, int H, int P, const float* xin, int xr0, int xrows, int Lx,
                        const float* yin, int yr0, int yrows, int Ly, int nfam, int xr0b, int Lxb, int yr0b, int Lyb,
                        const F2Ws& ws, float* out, hipStream_t st) {
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    auto proj = [&](GemmProblem& p, const float* in, int r0, int rows, const float* Wm, const float* b, int coloff) {
        p.A = in + (size_t)r0 * C; p.W = Wm; p.bias = b; p.out = ws.qkv + (size_t)r0 * 3 * C + coloff;
        p.rows = rows; p.ncols = C; p.K = C; p.K1 = C; p.lda = C; p.ldo = 3 * C; p.epi = EPI_NONE; p.scale = 1.f;
    };
    proj(g.p[0], xin, xr0, xrows, W.q_w, W.q_b, 0);
    proj(g.p[1], yin, yr0, yrows, W.k_w, W.k_b, C);
    proj(g.p[2], yin, yr0, yrows, W.v_w, W.v_b, 2 * C);
    g.n = 3;
    int rc = launch_gemm(g, st);
    if (rc) return rc;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = ws.qkv; a.k = ws.qkv + C; a.v = ws.qkv + 2 * C; a.out = ws.att;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C; a.H = H; a.d = C / H;
    a.nseg = P; a.q0 = xr0; a.qstride = Lx; a.Lq = Lx; a.k0 = yr0; a.kstride = Ly; a.Lk = Ly;
    if (nfam == 2) { a.nseg2 = P; a.q0b = xr0b; a.qstrideb = Lxb; a.Lqb = Lxb; a.k0b = yr0b; a.kstrideb = Lyb; a.Lkb = Lyb; }
    a.scale = 1.0f / sqrtf((float)(C / H));
    rc = launch_attention(a, st);
    if (rc) return rc;
    // z = norm(linear(h) + x)
    rc = gemm1(ws.att + (size_t)xr0 * C, C, W.lin_w, W.lin_b, ws.lin + (size_t)xr0 * C, C, xrows, C, C, EPI_NONE, 1.f, nullptr, st);
    if (rc) return rc;
    rc = launch_layernorm_postadd(ws.lin + (size_t)xr0 * C, C, W.norm1_w, W.norm1_b, xin + (size_t)xr0 * C, C, ws.z + (size_t)xr0 * C, C, xrows, C, st);
    if (rc) return rc;
    // out = norm(z + squeeze(relu(expand(z))))
    rc = gemm1(ws.z + (size_t)xr0 * C, C, W.expand_w, W.expand_b, ws.hid + (size_t)xr0 * 2 * C, 2 * C, xrows, 2 * C, C, EPI_RELU, 1.f, nullptr, st);
    if (rc) return rc;
    rc = gemm1(ws.hid + (size_t)xr0 * 2 * C, 2 * C, W.squeeze_w, W.squeeze_b, ws.f + (size_t)xr0 * C, C, xrows, C, 2 * C, EPI_NONE, 1.f, nullptr, st);
    if (rc) return rc;
    return launch_layernorm_postadd(ws.f + (size_t)xr0 * C, C, W.norm2_w, W.norm2_b, ws.z + (size_t)xr0 * C, C, out + (size_t)xr0 * C, C, xrows, C, st);
}

}  // namespace dr

using namespace dr;

extern "C" {

size_t dr_denoise_loop_2d3d_workspace_bytes(const dr_loop2d3d_config* cfg, int P, int N, int M) {
    if (!cfg || P < 1 || N < 1 || M < 1) return 0;
    Carver2 c(nullptr);
    F2Ws w;
    return F2Ws::carve(c, w, *cfg, P, N, M);
}

int dr_denoise_loop_2d3d(const dr_loop2d3d_config* cfg, const dr_fusion_weights* w, int P, int N, int M, const float* img_feats,
                         const float* img_dino, const float* img_pixels, const float* pcd_feats, const float* s_pcd,
                         const float* t_pcd_da, const uint8_t* src_mask, const uint8_t* tgt_mask, const uint8_t* tgt_mask_da,
                         const float* x_T, double* conf, double* x_final, int64_t* matches, int32_t* match_count, float* img_out,
                         float* pcd_out, const dr_loop_trace* trace, void* workspace, size_t workspace_bytes, void* stream) {
    if (!cfg || !w || P < 1 || N < 1 || M < 1 || !w->layers || !img_feats || !img_dino || !img_pixels || !pcd_feats || !s_pcd || !conf)
        return DR_EINVAL;
    if (cfg->C % cfg->H || (cfg->C / cfg->H) % 4 || cfg->img_dim % 4 || cfg->dino_dim % 4 || cfg->pcd_dim % 4 || cfg->steps < 0)
        return DR_EINVAL;
    const bool masked = src_mask != nullptr;
    if (masked != (tgt_mask != nullptr) || masked != (tgt_mask_da != nullptr)) return DR_EINVAL;
    if (cfg->steps > 0 && (!x_T || !t_pcd_da || !cfg->h_alphas_cumprod || !cfg->h_times)) return DR_EINVAL;
    if ((matches == nullptr) != (match_count == nullptr)) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_denoise_loop_2d3d_workspace_bytes(cfg, P, N, M)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver2 c(workspace);
    F2Ws L;
    F2Ws::carve(c, L, *cfg, P, N, M);
    const int C = cfg->C, H = cfg->H, PM = P * M, PN = P * N, T = PM + PN;
    const size_t NM = (size_t)P * N * M;
    const int strict = (cfg->flags & DR_LOOP_STRICT_F64) ? DR_SK_STRICT : 0;
    const int mflag = masked ? DR_SK_APPLY_MASK : 0;
    int rc;

    // ---- step-invariant tokens (fusion_module.py:84-92): image tokens incl. their 2-D embedding, point tokens w/o embedding
    rc = gemm1(img_feats, cfg->img_dim, w->img_in_w, w->img_in_b, L.cat, 2 * C, PM, C, cfg->img_dim, EPI_RELU, 1.f, nullptr, st);
    if (rc) return rc;
    rc = gemm1(img_dino, cfg->dino_dim, w->dino_w, w->dino_b, L.cat + C, 2 * C, PM, C, cfg->dino_dim, EPI_RELU, 1.f, nullptr, st);
    if (rc) return rc;
    rc = gemm1(L.cat, 2 * C, w->all_w, w->all_b, L.tok0, C, PM, C, 2 * C, EPI_NONE, 1.f, nullptr, st);
    if (rc) return rc;
    rc = launch_fourier2d(img_pixels, PM, 10, L.emb2, 44, st);
    if (rc) return rc;
    rc = gemm1(L.emb2, 44, w->img_emb_w, w->img_emb_b, L.tok0, C, PM, C, 44, EPI_NONE, 1.f, L.tok0, st);     // tokens += embedding
    if (rc) return rc;
    // point tokens WITHOUT their embedding are step-invariant too: parked in L.proj's point rows (the per-step
    // embedding projection adds them back through the GEMM's addend)
    rc = gemm1(pcd_feats, cfg->pcd_dim, w->pcd_in_w, w->pcd_in_b, L.proj + (size_t)PM * C, C, PN, C, cfg->pcd_dim, EPI_NONE, 1.f, nullptr, st);
    if (rc) return rc;

    auto evaluate = [&](const float* Rf, const float* tf) -> int {
        // point tokens = base + pcd_emb_proj(Fourier(warped - mean))  (fusion_module.py:55-59, 93-94)
        int r = launch_fourier3d(s_pcd, P, N, Rf, tf, 10, L.emb3, 64, L.warped, st);
        if (r) return r;
        r = gemm1(L.emb3, 64, w->pcd_emb_w, w->pcd_emb_b, L.tok0 + (size_t)PM * C, C, PN, C, 64, EPI_NONE, 1.f, L.proj + (size_t)PM * C, st);
        if (r) return r;
        const float* cur = L.tok0;
        float* bufs[2] = {L.ta, L.tb};
        int which = 0;
        for (int l = 0; l < cfg->n_layers; ++l) {
            float* nxt = bufs[which];
            if (l % 2 == 0) {   // self: image tokens and point tokens, same weights, one launch family each
                r = fusion_layer(w->layers[l], C, H, P, cur, 0, T, M, cur, 0, T, M, 2, PM, N, PM, N, L, nxt, st);
                if (r) return r;
            } else {            // cross: image <- points, then points <- UPDATED image (fusion_module.py:101-102)
                r = fusion_layer(w->layers[l], C, H, P, cur, 0, PM, M, cur, PM, PN, N, 1, 0, 0, 0, 0, L, nxt, st);
                if (r) return r;
                r = fusion_layer(w->layers[l], C, H, P, cur, PM, PN, N, nxt, 0, PM, M, 1, 0, 0, 0, 0, L, nxt, st);
                if (r) return r;
            }
            cur = nxt;
            which ^= 1;
        }
        r = gemm1(cur, C, w->out_w, w->out_b, L.feat, C, T, C, C, EPI_NONE, 1.f, nullptr, st);
        if (r) return r;
        // matching head: src_proj on both sides (Q1), / sqrt(C), sim[p] = pcd_p img_p^T  (EXP/matching.py:100-125)
        r = gemm1(L.feat, C, w->src_proj, nullptr, L.z, C, T, C, C, EPI_NONE, 1.0f / sqrtf((float)C), nullptr, st);
        if (r) return r;
        for (int p0 = 0; p0 < P; p0 += 4) {
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            g.n = (P - p0) < 4 ? (P - p0) : 4;
            for (int i = 0; i < g.n; ++i) {
                GemmProblem& q = g.p[i];
                const int pr = p0 + i;
                q.A = L.z + ((size_t)PM + (size_t)pr * N) * C; q.W = L.z + (size_t)pr * M * C;
                q.out = L.sim + (size_t)pr * N * M; q.rows = N; q.ncols = M; q.K = C; q.K1 = C; q.lda = C; q.ldo = M;
                q.epi = EPI_NONE; q.scale = 1.f;
            }
            r = launch_gemm(g, st);
            if (r) return r;
        }
        return sinkhorn_f32(P, N, M, L.sim, src_mask, tgt_mask, w->bin_score, cfg->sk_iters, DR_SK_OUT_CONF | mflag, L.x0, L.skws,
                            L.skws_bytes, st);
    };
    if (cfg->steps == 0) {          // component mode: one evaluation on the points as given
        rc = evaluate(nullptr, nullptr);
        if (rc) return rc;
        if (img_out) DR_HIP_CHECK(hipMemcpyAsync(img_out, L.feat, (size_t)PM * C * 4, hipMemcpyDeviceToDevice, st));
        if (pcd_out) DR_HIP_CHECK(hipMemcpyAsync(pcd_out, L.feat + (size_t)PM * C, (size_t)PN * C * 4, hipMemcpyDeviceToDevice, st));
        return launch_f32_to_f64(L.x0, conf, NM, st);
    }

    rc = launch_f32_to_f64(x_T, L.x, NM, st);
    if (rc) return rc;
    const double* ac = cfg->h_alphas_cumprod;
    for (int k = 0; k < cfg->steps; ++k) {
        const int tcur = cfg->h_times[k], tnext = cfg->h_times[k + 1];
        // warp from the noisy matrix: masks (src, tgt_da), no min-shift (EXP/model.py:830-846)
        rc = sinkhorn_f64(P, N, M, L.x, nullptr, src_mask, tgt_mask_da, w->bin_score, cfg->sk_iters,
                          DR_SK_OUT_CONF | DR_SK_OUT_F32 | mflag | (k > 0 ? strict : 0), L.wconf, L.skws, L.skws_bytes, st);
        if (rc) return rc;
        rc = launch_procrustes(L.wconf, s_pcd, t_pcd_da, src_mask, tgt_mask_da, P, N, M, 1, cfg->sample_rate, cfg->max_condition_num,
                               L.R, L.t, L.Rf, L.tf, L.cond, L.ok, nullptr, st, L.pws, L.pws_bytes);
        if (rc) return rc;
        if (trace && trace->R_forwd) DR_HIP_CHECK(hipMemcpyAsync(trace->R_forwd + (size_t)k * P * 9, L.Rf, (size_t)P * 36, hipMemcpyDeviceToDevice, st));
        if (trace && trace->t_forwd) DR_HIP_CHECK(hipMemcpyAsync(trace->t_forwd + (size_t)k * P * 3, L.tf, (size_t)P * 12, hipMemcpyDeviceToDevice, st));
        if (trace && trace->cond) DR_HIP_CHECK(hipMemcpyAsync(trace->cond + (size_t)k * P, L.cond, (size_t)P * 8, hipMemcpyDeviceToDevice, st));
        rc = evaluate(L.Rf, L.tf);
        if (rc) return rc;
        if (trace && trace->x0) DR_HIP_CHECK(hipMemcpyAsync(trace->x0 + (size_t)k * NM, L.x0, NM * 4, hipMemcpyDeviceToDevice, st));
        const double a = ac[tcur], an = ac[tnext];
        DdimArgs d;
        d.x = L.x; d.x0 = L.x0; d.shift = nullptr; d.noise = nullptr;
        d.src_mask = src_mask; d.tgt_mask = tgt_mask_da;          // the in-place fill of :832-834 persists in x
        d.N = N; d.M = M; d.first_step = (k == 0);
        d.sra = sqrt(1.0 / a); d.srm1 = sqrt(1.0 / a - 1.0);
        d.sigma = 1.0 * sqrt((1.0 - a / an) * (1.0 - an) / (1.0 - a));
        d.c = sqrt(1.0 - an - d.sigma * d.sigma);
        d.sqrt_an = (float)sqrt(an);
        rc = launch_ddim(d, P, st);
        if (rc) return rc;
    }
    if (x_final) DR_HIP_CHECK(hipMemcpyAsync(x_final, L.x, NM * 8, hipMemcpyDeviceToDevice, st));
    // read-out: no min-shift, masks (src, tgt) (EXP/model.py:681-694)
    rc = sinkhorn_f64(P, N, M, L.x, nullptr, src_mask, tgt_mask, w->bin_score, cfg->sk_iters, DR_SK_OUT_CONF | mflag | strict, conf,
                      L.skws, L.skws_bytes, st);
    if (rc) return rc;
    // (the steps' x0 tile is free by now; the row-block arg-maxima need < N M floats)
    if (matches) rc = launch_top1_union<double>(conf, P, N, M, (long long*)matches, match_count, st, nullptr, nullptr, L.x0, NM * 4);
    return rc;
}

}  // extern "C"
