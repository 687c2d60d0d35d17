// loop2d3d.hip -- orchestration of the 2D-3D variant: CrossModalFusionModule denoiser,
// position-free matching head, reverse sampling of MATR2D3D.forward.  Same kernels as loop.hip
// (GEMM with bias / addend epilogues, attention with d_head = 64, post-LN LayerNorm, Sinkhorn,
// Procrustes, DDIM), enqueued on one stream without host synchronisation.
//
// Token layout: image patches of all P pairs (pair p at row p*M), then point nodes (P*M + p*N).
//
// Round 4: calls of >= 4096 token rows (two cfg5 pairs or more) run the layer on fp16 hi / lo plane images (pgemm.h) like
// dr_denoise_loop: the biased q | k | v projections write the attention kernel's operand images, attention_planes_kernel (d = 64)
// writes the output projection's operand, z = LayerNorm(linear(h) + x) and out = LayerNorm(z + squeeze(relu(expand(z)))) are
// epilogue modes of the plane GEMM (bias, post-add LayerNorm, 256-column geometry) -- five launches per layer call instead of
// eight, no fp32 round trip of q / k / v / h / hidden.  The image half of layer 0 (a self layer on step-invariant tokens) is
// evaluated once per call.
#include "kernels.h"
#include "pgemm.h"
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

// ---- plane path ---------------------------------------------------------------------------------------------------------------
struct P2Layer {
    PgW qkv, lin, expand, squeeze;          // q | k | v: 3 blocks of C rows; expand: 2 blocks
    const float *qkv_b;                     // [3 C] the three biases back to back (a PgProblem's bias is indexed nb * C + col)
    const float *qkv_bmax, *lin_bmax, *exp_bmax, *sq_bmax;   // [3] / [1] / [2] / [1]
    const float *lnB1, *lnB2;
};
struct Prepack2 {
    static constexpr int MAXL = 16;
    P2Layer L[MAXL];
    PgW out, head;
    const float* out_bmax;
    static bool supported(const dr_loop2d3d_config& cfg) {
        if (cfg.n_layers > MAXL || cfg.C % cfg.H) return false;
        const int d = cfg.C / cfg.H;
        // heads start at a k-chunk without padding (d = 64): biases and images then share the nn.Linear's column order
        return pgemm_shape_ok(cfg.C) && (d == 64 || d == 112 || d == 144) && cfg.C / 16 >= 4;
    }
    static size_t carve(void* buf, const dr_loop2d3d_config& cfg, Prepack2* pp) {
        Carver2 c(buf);
        const int C = cfg.C, nC = C / 16;
        auto take = [&](int nblk, int nct, PgW* v) {
            char* p = c.take<char>(pgemm_weight_bytes(C, nblk, nct));
            if (pp && buf) pgemm_weight_view(p, C, nblk, nct, v);
        };
        for (int l = 0; l < cfg.n_layers; ++l) {
            P2Layer* L = pp ? &pp->L[l] : nullptr;
            take(3, nC, L ? &L->qkv : nullptr);
            take(1, nC, L ? &L->lin : nullptr);
            take(2, nC, L ? &L->expand : nullptr);
            take(1, 2 * nC, L ? &L->squeeze : nullptr);
            float* b = c.take<float>(3 * (size_t)C + 16);
            if (L && buf) {
                L->qkv_b = b; L->qkv_bmax = b + 3 * C; L->lin_bmax = b + 3 * C + 3; L->exp_bmax = b + 3 * C + 4; L->sq_bmax = b + 3 * C + 6;
                L->lnB1 = b + 3 * C + 7; L->lnB2 = b + 3 * C + 8;
            }
        }
        take(1, nC, pp ? &pp->out : nullptr);
        take(1, nC, pp ? &pp->head : nullptr);
        float* ob = c.take<float>(4);
        if (pp && buf) pp->out_bmax = ob;
        return c.off + 256;
    }
    static int fill(void* buf, const dr_loop2d3d_config& cfg, const dr_fusion_weights& W, hipStream_t st) {
        Prepack2 pp;
        carve(buf, cfg, &pp);
        const int C = cfg.C;
        for (int l = 0; l < cfg.n_layers; ++l) {
            const dr_fusion_layer_weights& w = W.layers[l];
            const P2Layer& L = pp.L[l];
            int rc = pgemm_pack_weights_block(w.q_w, C, C, C, C, L.qkv, 0, st);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.k_w, C, C, C, C, L.qkv, 1, st);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.v_w, C, C, C, C, L.qkv, 2, st);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.lin_w, C, C, C, C, L.lin, 0, st);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.expand_w, C, C, C, C, L.expand, 0, st);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.expand_w + (size_t)C * C, C, C, C, C, L.expand, 1, st);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.squeeze_w, C, 2 * C, 2 * C, 2 * C, L.squeeze, 0, st);
            if (rc) return rc;
            float* qb = const_cast<float*>(L.qkv_b);
            DR_HIP_CHECK(hipMemcpyAsync(qb, w.q_b, (size_t)C * 4, hipMemcpyDeviceToDevice, st));
            DR_HIP_CHECK(hipMemcpyAsync(qb + C, w.k_b, (size_t)C * 4, hipMemcpyDeviceToDevice, st));
            DR_HIP_CHECK(hipMemcpyAsync(qb + 2 * C, w.v_b, (size_t)C * 4, hipMemcpyDeviceToDevice, st));
            rc = launch_absmax_blocks(qb, 3, C, const_cast<float*>(L.qkv_bmax), st);
            if (rc == DR_OK) rc = launch_absmax_blocks(w.lin_b, 1, C, const_cast<float*>(L.lin_bmax), st);
            if (rc == DR_OK) rc = launch_absmax_blocks(w.expand_b, 2, C, const_cast<float*>(L.exp_bmax), st);
            if (rc == DR_OK) rc = launch_absmax_blocks(w.squeeze_b, 1, C, const_cast<float*>(L.sq_bmax), st);
            if (rc == DR_OK) rc = launch_ln_bound(w.norm1_w, w.norm1_b, C, const_cast<float*>(L.lnB1), st);
            if (rc == DR_OK) rc = launch_ln_bound(w.norm2_w, w.norm2_b, C, const_cast<float*>(L.lnB2), st);
            if (rc) return rc;
        }
        int rc = pgemm_pack_weights_block(W.out_w, C, C, C, C, pp.out, 0, st);
        if (rc == DR_OK) rc = pgemm_pack_weights_block(W.src_proj, C, C, C, C, pp.head, 0, st);
        if (rc == DR_OK) rc = launch_absmax_blocks(W.out_b, 1, C, const_cast<float*>(pp.out_bmax), st);
        return rc;
    }
};

// a token tensor of the plane path: fp32 rows [T, C], plane image (image part, then point part, each padded to 128 rows), bounds [T]
struct Tok2 { float* f32; char* img; float* bnd; };
struct Planes2 {
    bool on;
    Tok2 tok0, ta, tb, l0, z;
    char *qkv_img, *att_img, *hid_img, *feat_img;
    float *qkv_bnd, *att_bnd, *hid_bnd, *feat_bnd, *grp_x;
    size_t qkv_stride, side_C, side_hid;    // bytes from the q image to the k image; offset of the point part in an image of K = C / 2C
    void* own_pack;
    static size_t img_bytes(int PM, int PN, int K) { return plane_image_bytes(PM, K) + plane_image_bytes(PN, K); }
    static void carve(Carver2& c, Planes2& w, const dr_loop2d3d_config& cfg, int P, int N, int M) {
        const int C = cfg.C, PM = P * M, PN = P * N, T = PM + PN;
        w.on = env_knob("DR_PLANES", 1) && Prepack2::supported(cfg) && T >= env_knob("DR_PLANES_MIN_ROWS", 4096);
        if (cfg.flags & DR_LOOP_PLANES_FORCE) w.on = Prepack2::supported(cfg);
        if (cfg.flags & DR_LOOP_PLANES_OFF) w.on = false;
        if (!w.on) return;
        w.side_C = plane_image_bytes(PM, C); w.side_hid = plane_image_bytes(PM, 2 * C);
        Tok2* toks[5] = {&w.tok0, &w.ta, &w.tb, &w.l0, &w.z};
        for (Tok2* t : toks) { t->f32 = nullptr; t->img = c.take<char>(img_bytes(PM, PN, C)); t->bnd = c.take<float>(T); }
        w.l0.f32 = c.take<float>((size_t)T * C);
        w.qkv_stride = (img_bytes(PM, PN, C) + 255) & ~(size_t)255;
        w.qkv_img = c.take<char>(3 * w.qkv_stride); w.qkv_bnd = c.take<float>(3 * (size_t)T);
        w.att_img = c.take<char>(img_bytes(PM, PN, C)); w.att_bnd = c.take<float>(T);
        w.hid_img = c.take<char>(img_bytes(PM, PN, 2 * C)); w.hid_bnd = c.take<float>(T);
        w.feat_img = c.take<char>(img_bytes(PM, PN, C)); w.feat_bnd = c.take<float>(T);
        w.grp_x = c.take<float>(2 * (size_t)P);
        w.own_pack = c.take<char>(Prepack2::carve(nullptr, cfg, nullptr));
    }
};

struct F2Ws {
    float *tok0, *ta, *tb, *qkv, *att, *lin, *z, *hid, *f, *feat, *proj, *sim, *x0, *wconf, *cat, *emb2, *emb3, *warped;
    float *R, *t, *Rf, *tf;
    double *x, *cond;
    int* ok;
    void* skws; size_t skws_bytes;
    void* pws; size_t pws_bytes;
    Planes2 pl;
    unsigned* status;                        // the call's own sticky status word (dr_denoise_loop_status): zeroed when a call starts
    static size_t carve(Carver2& c, F2Ws& w, const dr_loop2d3d_config& cfg, int P, int N, int M) {
        const size_t T = (size_t)P * (N + M), C = cfg.C, NM = (size_t)P * N * M, PM = (size_t)P * M, PN = (size_t)P * N;
        w.status = c.take<unsigned>(4);      // (first, as in the 3D / 4D loop's workspace: dr_denoise_loop_status reads either)
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
        Planes2::carve(c, w.pl, cfg, P, N, M);
        if (w.pl.on) { w.pl.tok0.f32 = w.tok0; w.pl.ta.f32 = w.ta; w.pl.tb.f32 = w.tb; w.pl.z.f32 = w.z; }
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

// ---- one vision3d TransformerLayer call on plane images: five launches ---------------------------------------------------------
enum { SIDE_IMG = 1, SIDE_PCD = 2, SIDE_BOTH2 = 3 };
struct Fam2 { int q0, Lq, k0, Lk; };
struct P2Ctx { const Prepack2* pp; const Planes2* pw; int C, H, P, N, M; int attn_f16; };
static PgW pgw_blocks2(const PgW& v, int b0, int C) {
    PgW r = v;
    r.img += (size_t)b0 * v.nct * pgemm_bn(C) * 64; r.cinv += (size_t)b0 * pgemm_bn(C); r.wnorm += b0;
    return r;
}
// x rows of side(s) xs of `xin` attend the rows of side(s) ys of `yin`; out gets the x sides (fp32 rows + image + bounds)
static int fusion_layer_planes(const P2Ctx& X, const dr_fusion_layer_weights& W, int l, const Tok2& xin, int xs, const Tok2& yin, int ys,
                               const Tok2& out, const Fam2& f1, const Fam2* f2, hipStream_t st) {
    const int C = X.C, H = X.H, PM = X.P * X.M, PN = X.P * X.N, T = PM + PN, nC = C / 16, d = C / H;
    const P2Layer& L = X.pp->L[l];
    const Planes2& pw = *X.pw;
    auto r0 = [&](int side) { return side == SIDE_PCD ? PM : 0; };
    auto nrows = [&](int side) { return side == SIDE_PCD ? PN : PM; };
    auto per_pair = [&](int side) { return side == SIDE_PCD ? X.N : X.M; };
    auto at = [&](char* img, size_t side_off, int side) { return img + (side == SIDE_PCD ? side_off : 0); };
    PgBatch g;
    auto reset = [&]() { memset(&g, 0, sizeof(g)); };
    auto add = [&]() -> PgProblem& { return g.p[g.n++]; };
    auto for_sides = [&](int mask, auto fn) { for (int side = 1; side <= 2; ++side) if (mask & side) fn(side); };
    int rc = DR_OK;
    // bound of the keys' source rows per group (pair x side): all keys of a group share one scale in the k and v images
    // (taken inside the projection's kernel when a group is a whole number of workgroups: pgemm.h)
    const bool grp_inline = X.N % 128 == 0 && X.M % 128 == 0 && env_knob("DR_LOOP_GRP_INLINE", 1) != 0;
    for (int side = 1; side <= 2 && rc == DR_OK && !grp_inline; ++side)
        if (ys & side) rc = launch_group_max(yin.bnd + r0(side), X.P, per_pair(side), pw.grp_x + (side == SIDE_PCD ? X.P : 0), st);
    if (rc) return rc;
    // ---- q | k | v = x W^T + b -> three plane images (head h at k = h d), no rotary (vision3d/layers/transformer.py:96-104)
    auto proj = [&](const Tok2& tin, int side, int b0, int nblk, int grpm) {
        PgProblem& p = add();
        p.A0 = at(tin.img, pw.side_C, side); p.bnd0 = tin.bnd + r0(side); p.nc0 = nC;
        p.W = pgw_blocks2(L.qkv, b0, C); p.nblk = nblk; p.rows = nrows(side); p.C = C; p.k_alg = C; p.mode = PG_PLANES; p.scale = 1.f;
        p.bias = L.qkv_b + (size_t)b0 * C; p.bias_max = L.qkv_bmax + b0;
        p.pimg = at(pw.qkv_img + (size_t)b0 * pw.qkv_stride, pw.side_C, side); p.p_nct = nC; p.pbnd = pw.qkv_bnd + (size_t)b0 * T + r0(side);
        p.pimg_blk_stride = (long long)pw.qkv_stride; p.pbnd_blk_stride = T;
        p.grp_bnd = grp_inline ? nullptr : pw.grp_x; p.grp_mask = grpm; p.grp_first = side == SIDE_PCD ? X.P : 0; p.grp_rows = per_pair(side);
    };
    reset();
    if (xs == ys && xin.img == yin.img) {
        for_sides(xs, [&](int side) { proj(xin, side, 0, 3, 6); });
    } else {
        for_sides(xs, [&](int side) { proj(xin, side, 0, 1, 0); });
        for_sides(ys, [&](int side) { proj(yin, side, 1, 2, 3); });
    }
    rc = launch_pgemm(g, st);
    if (rc) return rc;
    // ---- attention on the images -> image of the heads' outputs
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.H = H; a.d = d;
    a.nseg = X.P; a.q0 = f1.q0; a.qstride = f1.Lq; a.Lq = f1.Lq; a.k0 = f1.k0; a.kstride = f1.Lk; a.Lk = f1.Lk;
    if (f2) { a.nseg2 = X.P; a.q0b = f2->q0; a.qstrideb = f2->Lq; a.Lqb = f2->Lq; a.k0b = f2->k0; a.kstrideb = f2->Lk; a.Lkb = f2->Lk; }
    a.scale = 1.0f / sqrtf((float)d);
    a.pimg[0] = pw.att_img; a.pimg[1] = pw.att_img + pw.side_C; a.p_split = PM; a.p_nct = nC; a.p_dp = d; a.pbnd = pw.att_bnd;
    a.qimg[0] = pw.qkv_img; a.qimg[1] = pw.qkv_img + pw.side_C;
    a.kimg[0] = pw.qkv_img + pw.qkv_stride; a.kimg[1] = a.kimg[0] + pw.side_C;
    a.vimg[0] = pw.qkv_img + 2 * pw.qkv_stride; a.vimg[1] = a.vimg[0] + pw.side_C;
    a.qbnd = pw.qkv_bnd; a.kgb = pw.qkv_bnd + T; a.vgb = pw.qkv_bnd + 2 * (size_t)T;
    a.f16_single = X.attn_f16;
    rc = launch_attention(a, st);
    if (rc) return rc;
    // ---- z = LayerNorm(linear(h) + b + x)   (transformer.py:188-196)
    reset();
    for_sides(xs, [&](int side) {
        PgProblem& p = add();
        p.A0 = at(pw.att_img, pw.side_C, side); p.bnd0 = pw.att_bnd + r0(side); p.nc0 = nC;
        p.W = L.lin; p.nblk = 1; p.rows = nrows(side); p.C = C; p.k_alg = C; p.mode = PG_LN; p.bias = W.lin_b;
        p.gamma = W.norm1_w; p.beta = W.norm1_b; p.lnB = L.lnB1;
        p.resid = xin.f32 + (size_t)r0(side) * C; p.ldr = C; p.ln_postadd = 1;
        p.out = pw.z.f32 + (size_t)r0(side) * C; p.ldo = C;
        p.pimg = at(pw.z.img, pw.side_C, side); p.p_nct = nC; p.pbnd = pw.z.bnd + r0(side);
    });
    rc = launch_pgemm(g, st);
    if (rc) return rc;
    // ---- hidden = relu(expand(z) + b)   (transformer.py:262-266)
    reset();
    for_sides(xs, [&](int side) {
        PgProblem& p = add();
        p.A0 = at(pw.z.img, pw.side_C, side); p.bnd0 = pw.z.bnd + r0(side); p.nc0 = nC;
        p.W = L.expand; p.nblk = 2; p.rows = nrows(side); p.C = C; p.k_alg = C; p.mode = PG_PLANES; p.relu = 1; p.scale = 1.f;
        p.bias = W.expand_b; p.bias_max = L.exp_bmax;
        p.pimg = at(pw.hid_img, pw.side_hid, side); p.p_nct = 2 * nC; p.pbnd = pw.hid_bnd + r0(side);
    });
    rc = launch_pgemm(g, st);
    if (rc) return rc;
    // ---- out = LayerNorm(z + squeeze(hidden) + b)   (transformer.py:267-271)
    reset();
    for_sides(xs, [&](int side) {
        PgProblem& p = add();
        p.A0 = at(pw.hid_img, pw.side_hid, side); p.bnd0 = pw.hid_bnd + r0(side); p.nc0 = 2 * nC;
        p.W = L.squeeze; p.nblk = 1; p.rows = nrows(side); p.C = C; p.k_alg = 2 * C; p.mode = PG_LN; p.bias = W.squeeze_b;
        p.gamma = W.norm2_w; p.beta = W.norm2_b; p.lnB = L.lnB2;
        p.resid = pw.z.f32 + (size_t)r0(side) * C; p.ldr = C; p.ln_postadd = 1;
        p.out = out.f32 + (size_t)r0(side) * C; p.ldo = C;
        p.pimg = at(out.img, pw.side_C, side); p.p_nct = nC; p.pbnd = out.bnd + r0(side);
    });
    return launch_pgemm(g, st);
}

}  // namespace dr

using namespace dr;

extern "C" {

size_t dr_loop2d3d_prepack_bytes(const dr_loop2d3d_config* cfg) {
    if (!cfg || cfg->C < 1 || cfg->H < 1 || !Prepack2::supported(*cfg)) return 0;
    return Prepack2::carve(nullptr, *cfg, nullptr);
}

int dr_loop2d3d_prepack(const dr_loop2d3d_config* cfg, const dr_fusion_weights* w, void* packed, size_t packed_bytes, void* stream) {
    if (!cfg || !w || !w->layers || !w->out_w || !w->out_b || !w->src_proj || !packed || ((uintptr_t)packed & 255)) return DR_EINVAL;
    if (cfg->C < 1 || cfg->H < 1 || !Prepack2::supported(*cfg)) return DR_ENOSUP;
    if (packed_bytes < Prepack2::carve(nullptr, *cfg, nullptr)) return DR_EWORKSPACE;
    return Prepack2::fill(packed, *cfg, *w, (hipStream_t)stream);
}

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
    if (trace && (trace->force_R == nullptr) != (trace->force_t == nullptr)) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_denoise_loop_2d3d_workspace_bytes(cfg, P, N, M)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver2 c(workspace);
    F2Ws L;
    F2Ws::carve(c, L, *cfg, P, N, M);
    DR_HIP_CHECK(hipMemsetAsync(L.status, 0, 16, st));          // the status of THIS call (dr_denoise_loop_status)
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

    // ---- plane path, once per call: packed weights (the caller's, or packed now into the workspace), the image tokens' plane image
    // and the image half of layer 0 (a self layer: image tokens attend image tokens only -- step-invariant)
    Prepack2 pp;
    const bool want_feat = cfg->steps == 0;
    if (L.pl.on) {
        void* buf = const_cast<void*>(w->prepacked);
        if (!buf) {
            buf = L.pl.own_pack;
            rc = Prepack2::fill(buf, *cfg, *w, st);
            if (rc) return rc;
        }
        Prepack2::carve(buf, *cfg, &pp);
        rc = launch_planes_from_f32(L.tok0, C, PM, C, L.pl.tok0.img, L.pl.tok0.bnd, st);
        if (rc) return rc;
        const P2Ctx X{&pp, &L.pl, C, H, P, N, M, (cfg->flags & DR_LOOP_ATTN_F16) ? 1 : 0};
        const Fam2 self_i{0, M, 0, M};
        rc = fusion_layer_planes(X, w->layers[0], 0, L.pl.tok0, SIDE_IMG, L.pl.tok0, SIDE_IMG, L.pl.l0, self_i, nullptr, st);
        if (rc) return rc;
    }

    auto evaluate = [&](const float* Rf, const float* tf) -> int {
        // point tokens = base + pcd_emb_proj(Fourier(warped - mean))  (fusion_module.py:55-59, 93-94)
        int r = launch_fourier3d(s_pcd, P, N, Rf, tf, 10, L.emb3, 64, L.warped, st);
        if (r) return r;
        r = gemm1(L.emb3, 64, w->pcd_emb_w, w->pcd_emb_b, L.tok0 + (size_t)PM * C, C, PN, C, 64, EPI_NONE, 1.f, L.proj + (size_t)PM * C, st);
        if (r) return r;
        if (L.pl.on) {
            const Planes2& pw = L.pl;
            const P2Ctx X{&pp, &pw, C, H, P, N, M, (cfg->flags & DR_LOOP_ATTN_F16) ? 1 : 0};
            const Fam2 self_i{0, M, 0, M}, self_p{PM, N, PM, N}, cross_i{0, M, PM, N}, cross_p{PM, N, 0, M};
            // the point tokens of this step -> their part of the token image (row maxima as bounds)
            r = launch_planes_from_f32(L.tok0 + (size_t)PM * C, C, PN, C, pw.tok0.img + pw.side_C, pw.tok0.bnd + PM, st);
            if (r) return r;
            // layer 0 (self): the image half is step-invariant and sits in pw.l0 already (l0_image_half below); the point half joins it there
            r = fusion_layer_planes(X, w->layers[0], 0, pw.tok0, SIDE_PCD, pw.tok0, SIDE_PCD, pw.l0, self_p, nullptr, st);
            if (r) return r;
            const Tok2* cur = &pw.l0;
            const Tok2* bufs[2] = {&pw.ta, &pw.tb};
            int which = 0;
            for (int l = 1; l < cfg->n_layers; ++l) {
                const Tok2* nxt = bufs[which];
                if (l % 2 == 0) {
                    r = fusion_layer_planes(X, w->layers[l], l, *cur, SIDE_BOTH2, *cur, SIDE_BOTH2, *nxt, self_i, &self_p, st);
                    if (r) return r;
                } else {        // image <- points, then points <- UPDATED image (fusion_module.py:101-102)
                    r = fusion_layer_planes(X, w->layers[l], l, *cur, SIDE_IMG, *cur, SIDE_PCD, *nxt, cross_i, nullptr, st);
                    if (r) return r;
                    r = fusion_layer_planes(X, w->layers[l], l, *cur, SIDE_PCD, *nxt, SIDE_IMG, *nxt, cross_p, nullptr, st);
                    if (r) return r;
                }
                cur = nxt;
                which ^= 1;
            }
            // out_proj (+ bias) -> image (+ fp32 rows when the caller wants the features), then the matching head's src_proj on both
            // sides (Q1) / sqrt(C) -> fp32 rows for the similarity
            PgBatch g;
            memset(&g, 0, sizeof(g));
            for (int side = 1; side <= 2; ++side) {
                PgProblem& p = g.p[g.n++];
                const int r0 = side == SIDE_PCD ? PM : 0;
                p.A0 = cur->img + (side == SIDE_PCD ? pw.side_C : 0); p.bnd0 = cur->bnd + r0; p.nc0 = C / 16;
                p.W = pp.out; p.nblk = 1; p.rows = side == SIDE_PCD ? PN : PM; p.C = C; p.k_alg = C; p.mode = PG_PLANES; p.scale = 1.f;
                p.bias = w->out_b; p.bias_max = pp.out_bmax;
                p.pimg = pw.feat_img + (side == SIDE_PCD ? pw.side_C : 0); p.p_nct = C / 16; p.pbnd = pw.feat_bnd + r0;
                if (want_feat) { p.out = L.feat + (size_t)r0 * C; p.ldo = C; }
            }
            r = launch_pgemm(g, st);
            if (r) return r;
            memset(&g, 0, sizeof(g));
            for (int side = 1; side <= 2; ++side) {
                PgProblem& p = g.p[g.n++];
                const int r0 = side == SIDE_PCD ? PM : 0;
                p.A0 = pw.feat_img + (side == SIDE_PCD ? pw.side_C : 0); p.bnd0 = pw.feat_bnd + r0; p.nc0 = C / 16;
                p.W = pp.head; p.nblk = 1; p.rows = side == SIDE_PCD ? PN : PM; p.C = C; p.k_alg = C; p.mode = PG_F32;
                p.out = L.z + (size_t)r0 * C; p.ldo = C; p.scale = 1.0f / sqrtf((float)C);
            }
            r = launch_pgemm(g, st);
            if (r) return r;
            GemmBatch gs;
            memset(&gs, 0, sizeof(gs));
            GemmProblem& q = gs.p[0];
            q.A = L.z + (size_t)PM * C; q.W = L.z; q.out = L.sim; q.rows = N; q.ncols = M; q.K = C; q.K1 = C; q.lda = C; q.ldo = M;
            q.epi = EPI_NONE; q.scale = 1.f; q.nbatch = P; q.sA = (long long)N * C; q.sW = (long long)M * C; q.sO = (long long)N * M;
            gs.n = 1;
            r = launch_gemm(gs, st);
            if (r) return r;
            return sinkhorn_f32(P, N, M, L.sim, src_mask, tgt_mask, w->bin_score, cfg->sk_iters, DR_SK_OUT_CONF | mflag, L.x0, L.skws,
                                L.skws_bytes, st, L.status);
        }
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
                            L.skws_bytes, st, L.status);
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
        // teacher forcing (parity tests): this step starts from the caller's state, not from the loop's own
        if (trace && trace->force_x) DR_HIP_CHECK(hipMemcpyAsync(L.x, trace->force_x + (size_t)k * NM, NM * 8, hipMemcpyDeviceToDevice, st));
        // warp from the noisy matrix: masks (src, tgt_da), no min-shift (EXP/model.py:830-846)
        rc = sinkhorn_f64(P, N, M, L.x, nullptr, src_mask, tgt_mask_da, w->bin_score, cfg->sk_iters,
                          DR_SK_OUT_CONF | DR_SK_OUT_F32 | mflag | (k > 0 ? strict : 0), L.wconf, L.skws, L.skws_bytes, st, L.status);
        if (rc) return rc;
        int* tk = nullptr;
        if (trace && trace->topk_idx) {
            const size_t Kf = (size_t)(int)((float)(N > M ? N : M) * cfg->sample_rate);
            tk = trace->topk_idx + (size_t)k * P * Kf;
            DR_HIP_CHECK(hipMemsetAsync(tk, 0xff, (size_t)P * Kf * 4, st));
        }
        if (trace && trace->wconf) DR_HIP_CHECK(hipMemcpyAsync(trace->wconf + (size_t)k * NM, L.wconf, NM * 4, hipMemcpyDeviceToDevice, st));
        rc = launch_procrustes(L.wconf, s_pcd, t_pcd_da, src_mask, tgt_mask_da, P, N, M, 1, cfg->sample_rate, cfg->max_condition_num,
                               L.R, L.t, L.Rf, L.tf, L.cond, L.ok, tk, st, L.pws, L.pws_bytes);
        if (rc) return rc;
        if (trace && trace->R_forwd) DR_HIP_CHECK(hipMemcpyAsync(trace->R_forwd + (size_t)k * P * 9, L.Rf, (size_t)P * 36, hipMemcpyDeviceToDevice, st));
        if (trace && trace->t_forwd) DR_HIP_CHECK(hipMemcpyAsync(trace->t_forwd + (size_t)k * P * 3, L.tf, (size_t)P * 12, hipMemcpyDeviceToDevice, st));
        if (trace && trace->cond) DR_HIP_CHECK(hipMemcpyAsync(trace->cond + (size_t)k * P, L.cond, (size_t)P * 8, hipMemcpyDeviceToDevice, st));
        if (trace && trace->force_R) {           // teacher forcing: warp with the caller's pose (the fit above is traced all the same)
            DR_HIP_CHECK(hipMemcpyAsync(L.Rf, trace->force_R + (size_t)k * P * 9, (size_t)P * 36, hipMemcpyDeviceToDevice, st));
            DR_HIP_CHECK(hipMemcpyAsync(L.tf, trace->force_t + (size_t)k * P * 3, (size_t)P * 12, hipMemcpyDeviceToDevice, st));
        }
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
        if (trace && trace->x_next) DR_HIP_CHECK(hipMemcpyAsync(trace->x_next + (size_t)k * NM, L.x, NM * 8, hipMemcpyDeviceToDevice, st));
    }
    if (x_final) DR_HIP_CHECK(hipMemcpyAsync(x_final, L.x, NM * 8, hipMemcpyDeviceToDevice, st));
    // read-out: no min-shift, masks (src, tgt) (EXP/model.py:681-694)
    rc = sinkhorn_f64(P, N, M, L.x, nullptr, src_mask, tgt_mask, w->bin_score, cfg->sk_iters, DR_SK_OUT_CONF | mflag | strict, conf,
                      L.skws, L.skws_bytes, st, L.status);
    if (rc) return rc;
    // (the steps' x0 tile is free by now; the row-block arg-maxima need < N M floats)
    if (matches) rc = launch_top1_union<double>(conf, P, N, M, (long long*)matches, match_count, st, nullptr, nullptr, L.x0, NM * 4);
    return rc;
}

}  // extern "C"
