// loop.hip -- host-side orchestration of the reverse-diffusion matching loop and the public
// entry points of the individual ops.  Everything is enqueued on the caller's stream; there is no
// host synchronisation anywhere, so the whole loop can be captured in a HIP graph.
//
// Token layout: all src superpoints of all P pairs, then all tgt superpoints:
//   rows [0, P*N) = src (pair p at p*N), rows [P*N, P*(N+M)) = tgt (pair p at P*N + p*M).
// Linear layers / LayerNorm / rotary are row-wise, so the P pairs simply widen the GEMMs; only the
// attention, the N x M matrices and the Procrustes fit know about pair boundaries.
#include "kernels.h"
#include "pgemm.h"
#include <string.h>

namespace dr {

struct Carver {
    char* base;
    size_t off, cap;
    Carver(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
};

// buffers of one attention-layer evaluation over `T` token rows
struct LayerWs {
    float *qkv, *att, *mrg, *msg, *hid, *g2;
    size_t T;
    static size_t carve(Carver& c, LayerWs& w, size_t T, int C) {
        w.T = T;
        w.qkv = c.take<float>(T * 3 * C);
        w.att = c.take<float>(T * C);
        w.mrg = c.take<float>(T * C);
        w.msg = c.take<float>(T * C);
        w.hid = c.take<float>(T * 2 * C);
        w.g2 = c.take<float>(T * C);
        return c.off;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Plane-image path of the layer GEMMs (pgemm.h): every activation that feeds a GEMM lives as an fp16 hi / lo plane image
// written by its producer; LayerNorm happens in the epilogue of the GEMM in front of it; weights are packed once.
// ---------------------------------------------------------------------------------------------------------------------
struct PrepackLayer {
    PgW qkv, merge, mlp0, mlp2;     // q|k|v: 3 blocks; merge: k order padded per head; mlp0: 2 blocks
    const float *lnB1, *lnB2;       // device scalars: bound of norm1 / norm2 outputs
};
struct Prepack {
    static constexpr int MAXL = 16;
    PrepackLayer L[MAXL];
    PgW head;                        // src_proj
    int dp;                          // head dim rounded up to 16
    static bool supported(const dr_loop_config& cfg) {
        if (!(cfg.n_layers <= MAXL && pgemm_shape_ok(cfg.C) && cfg.C % cfg.H == 0 && (cfg.C / cfg.H) % 4 == 0)) return false;
        const int dp = (cfg.C / cfg.H + 15) / 16 * 16;                 // the head-padded q | k | v block must fit the column block too
        return pgemm_shape_ok(cfg.H * dp) && pgemm_bn(cfg.H * dp) == pgemm_bn(cfg.C) && (dp == 64 || dp == 112 || dp == 144);
    }
    static bool wide_layout(const dr_loop_config& cfg) { return pgemm_bn(cfg.C) > 448 && env_knob("DR_PG_WIDE", 1) != 0; }
    // lays the images out in `buf` (nullptr: size only) and returns the byte count
    static size_t carve(void* buf, const dr_loop_config& cfg, Prepack* pp) {
        Carver c(buf, (size_t)-1);
        const int C = cfg.C, d = C / cfg.H, dp = (d + 15) / 16 * 16, nC = C / 16;
        if (pp) pp->dp = dp;
        auto take = [&](int nblk, int nct, PgW* v) {
            char* p = c.take<char>(pgemm_weight_bytes(C, nblk, nct));
            if (pp && buf) pgemm_weight_view(p, C, nblk, nct, v);
        };
        // the launches without LayerNorm of the 576-column geometry (4DMatch) run on the wide-wave kernel: their weights in its layout
        const bool wide = wide_layout(cfg);
        auto takew = [&](int nblk, int nct, PgW* v) {
            if (!wide) return take(nblk, nct, v);
            char* p = c.take<char>(pgemm16w_weight_bytes(nblk, nct));
            if (pp && buf) pgemm16w_weight_view(p, nblk, nct, v);
        };
        for (int l = 0; l < cfg.n_layers; ++l) {
            PrepackLayer* L = pp ? &pp->L[l] : nullptr;
            takew(3, nC, L ? &L->qkv : nullptr);           // (C' = H dp output columns per block: same image size, C' <= BN)
            take(1, cfg.H * dp / 16, L ? &L->merge : nullptr);
            takew(2, 2 * nC, L ? &L->mlp0 : nullptr);
            take(1, 2 * nC, L ? &L->mlp2 : nullptr);
            float* b = c.take<float>(2);
            if (L && buf) { L->lnB1 = b; L->lnB2 = b + 1; }
        }
        takew(1, nC, pp ? &pp->head : nullptr);
        return c.off + 256;
    }
    static int fill(void* buf, const dr_loop_config& cfg, const dr_loop_weights& W, hipStream_t st) {
        Prepack pp;
        carve(buf, cfg, &pp);
        const int C = cfg.C, d = C / cfg.H;
        // q|k|v are three separate [C, C] tensors: pack them as three one-block images laid out back to back (the block stride
        // of a 3-block image is exactly one 1-block image minus its tail, so pack block by block into the 3-block view)
        for (int l = 0; l < cfg.n_layers; ++l) {
            const dr_layer_weights& w = W.layers[l];
            const PrepackLayer& L = pp.L[l];
            // q | k | v: output columns padded per head (d -> dp) so that the images the GEMM writes start every head at a k-chunk
            const int Cq = cfg.H * pp.dp;
            auto packw = [&](const float* Wm, int Cc, int K, int plen, int ppad, const PgW& v, int nb, int olen = 0, int opad = 0) {
                return v.sub == 2 ? pgemm16w_pack_weights_block(Wm, Cc, K, plen, ppad, v, nb, st, olen, opad)
                                  : pgemm_pack_weights_block(Wm, Cc, K, plen, ppad, v, nb, st, olen, opad);
            };
            int rc = packw(w.q_proj, Cq, C, C, C, L.qkv, 0, d, pp.dp);
            if (rc == DR_OK) rc = packw(w.k_proj, Cq, C, C, C, L.qkv, 1, d, pp.dp);
            if (rc == DR_OK) rc = packw(w.v_proj, Cq, C, C, C, L.qkv, 2, d, pp.dp);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.merge, C, C, d, pp.dp, L.merge, 0, st);
            if (rc == DR_OK) rc = packw(w.mlp0, C, 2 * C, 2 * C, 2 * C, L.mlp0, 0);
            if (rc == DR_OK) rc = packw(w.mlp0 + (size_t)C * 2 * C, C, 2 * C, 2 * C, 2 * C, L.mlp0, 1);
            if (rc == DR_OK) rc = pgemm_pack_weights_block(w.mlp2, C, 2 * C, 2 * C, 2 * C, L.mlp2, 0, st);
            if (rc == DR_OK) rc = launch_ln_bound(w.norm1_w, w.norm1_b, C, (float*)L.lnB1, st);
            if (rc == DR_OK) rc = launch_ln_bound(w.norm2_w, w.norm2_b, C, (float*)L.lnB2, st);
            if (rc) return rc;
        }
        return pp.head.sub == 2 ? pgemm16w_pack_weights_block(W.src_proj, C, C, C, C, pp.head, 0, st)
                                : pgemm_pack_weights_block(W.src_proj, C, C, C, C, pp.head, 0, st);
    }
};

// a token tensor of the plane path: fp32 rows [T, C] (the residual stream; may be null), plane image (src part, then tgt
// part, each padded to 128 rows) and per-row bounds [T]
struct Tok {
    float* f32; char* img; float* bnd;
};
struct PlanesWs {
    bool on;
    Tok feat0, fa, fb, tgt_l0;
    char *att_img, *msg_img, *hid_img;
    float *att_bnd, *msg_bnd, *hid_bnd;
    char *qkv_img, *kvc_img;                 // q | k | v images (three images of H dp columns, back to back); cached k | v of layer 1
    float *qkv_bnd, *kvc_bnd;                // [3][T] / [2][T]
    float* grp_x;                            // [2 P] bound of x per key group (src groups, then tgt groups)
    float* csT;                              // [T][C/2][2] the rotary tables interleaved (cos, sin)
    size_t qkv_stride;                       // bytes from the q image to the k image (= to the next: v)
    size_t side_C, side_att, side_hid;      // byte offset of the tgt part inside an image of K = C / H dp / 2C
    void* own_pack;                         // packed weights inside the workspace (used when the caller passed none)
    // KSPLIT exchange of the LayerNorm launches (pgemm.h): partial sums + arrival flags (zeroed by planes_begin), the launches counted since then
    float* xk_buf; unsigned* xk_flags; mutable unsigned xk_epoch; unsigned* status;
    static size_t img_bytes(int PN, int PM, int K) { return plane_image_bytes(PN, K) + plane_image_bytes(PM, K); }
    static void carve(Carver& c, PlanesWs& w, const dr_loop_config& cfg, int P, int N, int M) {
        const int C = cfg.C, PN = P * N, PM = P * M, T = PN + PM;
        // (crossover re-measured with the 64-row plane workgroups, tools/bench_planes_threshold.py: 256-point pairs 24.5 / 37.7 ms on the
        //  f32 kernels against 31.6 / 31.9 ms on the plane path at 2048 / 4096 token rows; 512-point 4D pairs 60.9 vs 62.0 ms at 4096)
        const int min_rows = env_knob("DR_PLANES_MIN_ROWS", 4096);
        const int enabled = env_knob("DR_PLANES", 1);
        w.on = enabled && Prepack::supported(cfg) && T >= min_rows;
        if (cfg.flags & DR_LOOP_PLANES_FORCE) w.on = Prepack::supported(cfg);
        if (cfg.flags & DR_LOOP_PLANES_OFF) w.on = false;
        if (!w.on) return;
        const int dp = (C / cfg.H + 15) / 16 * 16;
        w.side_C = plane_image_bytes(PN, C); w.side_att = plane_image_bytes(PN, cfg.H * dp); w.side_hid = plane_image_bytes(PN, 2 * C);
        Tok* toks[4] = {&w.feat0, &w.fa, &w.fb, &w.tgt_l0};
        for (Tok* t : toks) { t->f32 = nullptr; t->img = c.take<char>(img_bytes(PN, PM, C)); t->bnd = c.take<float>(T); }
        w.att_img = c.take<char>(img_bytes(PN, PM, cfg.H * dp)); w.att_bnd = c.take<float>(T);
        w.msg_img = c.take<char>(img_bytes(PN, PM, C)); w.msg_bnd = c.take<float>(T);
        w.hid_img = c.take<char>(img_bytes(PN, PM, 2 * C)); w.hid_bnd = c.take<float>(T);
        w.qkv_stride = (img_bytes(PN, PM, cfg.H * dp) + 255) & ~(size_t)255;
        w.qkv_img = c.take<char>(3 * w.qkv_stride); w.qkv_bnd = c.take<float>(3 * (size_t)T);
        w.kvc_img = c.take<char>(2 * w.qkv_stride); w.kvc_bnd = c.take<float>(2 * (size_t)T);
        w.grp_x = c.take<float>(2 * (size_t)P);
        w.csT = c.take<float>((size_t)T * C);
        w.own_pack = c.take<char>(Prepack::carve(nullptr, cfg, nullptr));
        // (only launches of at most half a chip of 64-row workgroups are split: 128 row blocks on 256 CUs)
        const bool xk = (T + 63) / 64 + 2 <= PG_XK_MAX_RB;
        w.xk_buf = xk ? c.take<float>(pgemm_xk_buf_bytes(pgemm_bn(C)) / 4) : nullptr;
        w.xk_flags = xk ? c.take<unsigned>(pgemm_xk_flag_bytes() / 4) : nullptr;
        w.xk_epoch = 0; w.status = nullptr;
    }
};

struct Family {   // P segments: queries rows q0 + p*Lq (+Lq) attend keys rows k0 + p*Lk (+Lk)
    int q0, Lq, k0, Lk;
};

// One GeometryAttentionLayer call (transformero.py:43-96) on token buffers.
//   xin rows [xr0, xr0 + xrows) are the queries/residual stream, yin rows [yr0, yr0 + yrows) the
//   source; out gets rows [xr0, xr0 + xrows).  fam2 may be null.
static int layer_call(const dr_layer_weights& W, int C, int H, int P, const float* xin, int xr0, int xrows,
                      const float* yin, int yr0, int yrows, const float* cosT, const float* sinT,
                      const uint8_t* tokmask, const Family& f1, const Family* f2, const LayerWs& ws, float* out,
                      hipStream_t st, const float* kv_cached = nullptr, float* kv_store = nullptr, const float* xq_in = nullptr,
                      const float* yk_in = nullptr) {
    // kv_cached: K|V of the source rows were projected earlier ([tokens, 2C], rotary applied to K): skip them.
    // kv_store : project ONLY K|V of the source rows into this buffer and return (used to fill the cache).
    // xq_in / yk_in: the inputs of the q / k projections where they are not x / y themselves (pe_type 'sinusoidal', transformero.py:50-57:
    //   q = W_q (x + pe_x), k = W_k (y + pe_y), v = W_v y); cosT == nullptr: no rotary code on q and k (sinusoidal, or the entangled form whose
    //   layers are called without a position code, transformero.py:246-252)
    const int halfC = C / 2, d = C / H;
    const bool rotary = cosT != nullptr;
    const float* const qin = xq_in ? xq_in : xin;
    const float* const kin = yk_in ? yk_in : yin;
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    auto proj = [&](GemmProblem& p, const float* in, int r0, int rows, const float* Wm, int coloff, bool rot) {
        p.A = in + (size_t)r0 * C; p.A2 = nullptr; p.W = Wm; p.out = ws.qkv + (size_t)r0 * 3 * C + coloff;
        p.rows = rows; p.ncols = C; p.K = C; p.K1 = C; p.lda = C; p.lda2 = 0; p.ldo = 3 * C;
        p.epi = (rot && rotary) ? EPI_ROTARY : EPI_NONE; p.rot_C = C; p.scale = 1.f;
        p.cosT = rotary ? cosT + (size_t)r0 * halfC : nullptr; p.sinT = rotary ? sinT + (size_t)r0 * halfC : nullptr;
    };
    int rc;
    if (kv_store) {
        proj(g.p[0], kin, yr0, yrows, W.k_proj, 0, true);
        proj(g.p[1], yin, yr0, yrows, W.v_proj, 0, false);
        g.p[0].out = kv_store + (size_t)yr0 * 2 * C; g.p[0].ldo = 2 * C;
        g.p[1].out = kv_store + (size_t)yr0 * 2 * C + C; g.p[1].ldo = 2 * C;
        g.n = 2;
        return launch_gemm(g, st);
    }
    proj(g.p[0], qin, xr0, xrows, W.q_proj, 0, true);
    g.n = 1;
    if (!kv_cached) {
        proj(g.p[1], kin, yr0, yrows, W.k_proj, C, true);
        proj(g.p[2], yin, yr0, yrows, W.v_proj, 2 * C, false);
        g.n = 3;
    }
    rc = launch_gemm(g, st);
    if (rc) return rc;

    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = ws.qkv; a.k = ws.qkv + C; a.v = ws.qkv + 2 * C; a.out = ws.att;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C; a.H = H; a.d = d;
    if (kv_cached) { a.k = kv_cached; a.v = kv_cached + C; a.ldk = a.ldv = 2 * C; }
    a.qmask = tokmask; a.kmask = tokmask;
    a.nseg = P; a.q0 = f1.q0; a.qstride = f1.Lq; a.Lq = f1.Lq; a.k0 = f1.k0; a.kstride = f1.Lk; a.Lk = f1.Lk;
    if (f2) {
        a.nseg2 = P; a.q0b = f2->q0; a.qstrideb = f2->Lq; a.Lqb = f2->Lq; a.k0b = f2->k0; a.kstrideb = f2->Lk; a.Lkb = f2->Lk;
    }
    a.scale = 1.0f / sqrtf((float)d);
    rc = launch_attention(a, st);
    if (rc) return rc;

    // message = norm1(merge(o))
    memset(&g, 0, sizeof(g));
    GemmProblem& m = g.p[0];
    m.A = ws.att + (size_t)xr0 * C; m.W = W.merge; m.out = ws.mrg + (size_t)xr0 * C;
    m.rows = xrows; m.ncols = C; m.K = C; m.K1 = C; m.lda = C; m.ldo = C; m.epi = EPI_NONE; m.scale = 1.f;
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    rc = launch_layernorm(ws.mrg + (size_t)xr0 * C, C, W.norm1_w, W.norm1_b, nullptr, 0, ws.msg + (size_t)xr0 * C, C, xrows, C, st);
    if (rc) return rc;
    // message = norm2(mlp(cat[x, message]))
    memset(&g, 0, sizeof(g));
    GemmProblem& h = g.p[0];
    h.A = xin + (size_t)xr0 * C; h.A2 = ws.msg + (size_t)xr0 * C; h.W = W.mlp0; h.out = ws.hid + (size_t)xr0 * 2 * C;
    h.rows = xrows; h.ncols = 2 * C; h.K = 2 * C; h.K1 = C; h.lda = C; h.lda2 = C; h.ldo = 2 * C; h.epi = EPI_RELU; h.scale = 1.f;
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    GemmProblem& o = g.p[0];
    o.A = ws.hid + (size_t)xr0 * 2 * C; o.W = W.mlp2; o.out = ws.g2 + (size_t)xr0 * C;
    o.rows = xrows; o.ncols = C; o.K = 2 * C; o.K1 = 2 * C; o.lda = 2 * C; o.ldo = C; o.epi = EPI_NONE; o.scale = 1.f;
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    // e = x + message
    return launch_layernorm(ws.g2 + (size_t)xr0 * C, C, W.norm2_w, W.norm2_b, xin + (size_t)xr0 * C, C, out + (size_t)xr0 * C, C,
                            xrows, C, st);
}


// ---- the plane path of one GeometryAttentionLayer call (transformero.py:43-96): five launches ---------------------------------
enum { SIDE_SRC = 1, SIDE_TGT = 2, SIDE_BOTH = 3 };
struct PlCtx {
    const Prepack* pp; const PlanesWs* pw; const LayerWs* lw;
    int C, H, P, N, M;
    const float *cosT, *sinT;
    const uint8_t* tokmask;
    int attn_f16;                            // DR_LOOP_ATTN_F16
};
static PgW pgw_blocks(const PgW& v, int b0, int C) {
    PgW r = v;
    if (v.sub == 2) { r.img += (size_t)b0 * v.nct * 576 * 64; r.cinv += (size_t)b0 * 576; r.wnorm += b0; return r; }   // (two sub-blocks of 288 rows per block)
    r.img += (size_t)b0 * v.nct * pgemm_bn(C) * 64; r.cinv += (size_t)b0 * pgemm_bn(C); r.wnorm += b0;
    return r;
}
// exchange region of a problem whose launch may split its k range over two workgroups (pgemm.h: xk_buf).  The buffer holds PG_XK_MAX_RB units --
// one per 64-row block of a LayerNorm launch / per 128 x 288 tile of a wide-wave launch; the problems of a launch take consecutive regions
// (`next`: units handed out so far in this launch); the caller advances pw.xk_epoch once per launch that got regions.
static void xk_assign(const PlanesWs& pw, int C, PgProblem& p, size_t& next, size_t units) {
    if (!pw.xk_buf || next + units > (size_t)PG_XK_MAX_RB) return;
    p.xk_buf = pw.xk_buf + next * (pgemm_xk_buf_bytes(pgemm_bn(C)) / 4 / PG_XK_MAX_RB); p.xk_flags = pw.xk_flags + next * 2; p.xk_cap = (int)units;
    p.xk_epoch = pw.xk_epoch + 1; p.xk_status = pw.status;
    next += units;
}
static size_t xk_wide_units(const PgProblem& p) { return (size_t)(p.rows + 127) / 128 * p.nblk * 2; }

static int layer_call_planes(const PlCtx& X, const dr_layer_weights& W, int l, const Tok& xin, int xs, const Tok& yin, int ys,
                             const Tok& out, const Family& f1, const Family* f2, hipStream_t st, const float* kv_cached = nullptr,
                             float* kv_store = nullptr) {
    const int C = X.C, H = X.H, PN = X.P * X.N, PM = X.P * X.M, halfC = C / 2, d = C / H, nC = C / 16, dp = X.pp->dp;
    const PrepackLayer& L = X.pp->L[l];
    const PlanesWs& pw = *X.pw;
    auto r0 = [&](int side) { return side == SIDE_TGT ? PN : 0; };
    auto nrows = [&](int side) { return side == SIDE_TGT ? PM : PN; };
    auto at = [&](char* img, size_t side_off, int side) { return img + (side == SIDE_TGT ? side_off : 0); };
    PgBatch g;
    size_t xk_next = 0;                   // exchange units handed to the problems of the launch being assembled
    auto reset = [&]() { memset(&g, 0, sizeof(g)); xk_next = 0; };
    auto add = [&]() -> PgProblem& { return g.p[g.n++]; };
    auto for_sides = [&](int mask, auto fn) { for (int side = 1; side <= 2; ++side) if (mask & side) fn(side); };
    // KSPLIT exchange region of a LayerNorm problem (launch_pgemm decides whether the launch is split): the tgt side's row blocks behind the src side's
    auto xk = [&](PgProblem& p, size_t units) { xk_assign(pw, C, p, xk_next, units); };
    int rc;
    bool rc_ok = true;

    // ---- q | k | v projections + rotary -> three plane images of H dp columns (head h at k = h dp): the attention kernel's
    // operands.  q keeps a scale per row; all keys of a pair's side share ONE scale (k, v blocks take the bound of the row's group).
    const int T = PN + PM, Cq = H * dp, nq = Cq / 16;
    const bool cached = kv_cached != nullptr;            // (the plane path keeps the cached K | V as images: kvc_img / kvc_bnd)
    char* const kv_img = kv_store ? pw.kvc_img : pw.qkv_img + pw.qkv_stride;
    float* const kv_bnd = kv_store ? pw.kvc_bnd : pw.qkv_bnd + T;
    // bound of the keys' source rows per group (pair x side): taken inside the projection's own kernel when a group is a whole number of
    // workgroups (the 480 five-microsecond launches of a 20-step loop were 2 - 4 % of it), by a kernel of its own otherwise
    const bool grp_inline = X.N % 128 == 0 && X.M % 128 == 0 && env_knob("DR_LOOP_GRP_INLINE", 1) != 0;
    if (!cached && !grp_inline) {
        for (int side = 1; side <= 2 && rc_ok; ++side)
            if (ys & side) rc_ok = launch_group_max(yin.bnd + r0(side), X.P, side == SIDE_TGT ? X.M : X.N, pw.grp_x + (side == SIDE_TGT ? X.P : 0), st) == DR_OK;
        if (!rc_ok) return DR_ELAUNCH;
    }
    auto proj = [&](const Tok& tin, int side, int b0, int nblk, char* img, float* bnd, int rotm, int grpm) {
        PgProblem& p = add();
        p.A0 = at(tin.img, pw.side_C, side); p.bnd0 = tin.bnd + r0(side); p.nc0 = nC;
        p.W = pgw_blocks(L.qkv, b0, C); p.nblk = nblk; p.rows = nrows(side); p.C = Cq; p.k_alg = C; p.mode = PG_PLANES;
        p.rot_mask = rotm; p.rot_C = C; p.rot_piece_len = d; p.rot_piece_pad = dp; p.scale = 1.f;
        p.cosT = X.cosT + (size_t)r0(side) * halfC; p.sinT = X.sinT + (size_t)r0(side) * halfC;
        p.csT = X.pw->csT + (size_t)r0(side) * halfC * 2;
        p.pimg = at(img, pw.side_att, side); p.p_nct = nq; p.pbnd = bnd + r0(side);
        p.pimg_blk_stride = (long long)pw.qkv_stride; p.pbnd_blk_stride = T;
        p.grp_bnd = grp_inline ? nullptr : pw.grp_x; p.grp_mask = grpm; p.grp_first = side == SIDE_TGT ? X.P : 0; p.grp_rows = side == SIDE_TGT ? X.M : X.N;
        if (p.W.sub == 2) xk(p, xk_wide_units(p));
    };
    reset();
    if (kv_store) {
        for_sides(ys, [&](int side) { proj(yin, side, 1, 2, kv_img, kv_bnd, 1, 3); });
        ++pw.xk_epoch;
        return launch_pgemm(g, st);
    }
    const bool self = xs == ys && xin.img == yin.img && !cached;
    if (self) {
        for_sides(xs, [&](int side) { proj(xin, side, 0, 3, pw.qkv_img, pw.qkv_bnd, 3, 6); });
    } else {
        for_sides(xs, [&](int side) { proj(xin, side, 0, 1, pw.qkv_img, pw.qkv_bnd, 1, 0); });
        if (!cached) for_sides(ys, [&](int side) { proj(yin, side, 1, 2, kv_img, kv_bnd, 1, 3); });
    }
    ++pw.xk_epoch;
    rc = launch_pgemm(g, st);
    if (rc) return rc;

    // ---- attention on the images -> plane image of the heads' outputs (head h at k = h dp)
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.H = H; a.d = d;
    a.qmask = X.tokmask; a.kmask = X.tokmask;
    a.nseg = X.P; a.q0 = f1.q0; a.qstride = f1.Lq; a.Lq = f1.Lq; a.k0 = f1.k0; a.kstride = f1.Lk; a.Lk = f1.Lk;
    if (f2) { a.nseg2 = X.P; a.q0b = f2->q0; a.qstrideb = f2->Lq; a.Lqb = f2->Lq; a.k0b = f2->k0; a.kstrideb = f2->Lk; a.Lkb = f2->Lk; }
    a.scale = 1.0f / sqrtf((float)d);
    a.pimg[0] = pw.att_img; a.pimg[1] = pw.att_img + pw.side_att; a.p_split = PN; a.p_nct = nq; a.p_dp = dp;
    a.pbnd = pw.att_bnd;
    {
        const char* kimg = cached ? pw.kvc_img : pw.qkv_img + pw.qkv_stride;
        const float* kb = cached ? pw.kvc_bnd : pw.qkv_bnd + T;
        a.qimg[0] = pw.qkv_img; a.qimg[1] = pw.qkv_img + pw.side_att;
        a.kimg[0] = kimg; a.kimg[1] = kimg + pw.side_att;
        a.vimg[0] = kimg + pw.qkv_stride; a.vimg[1] = kimg + pw.qkv_stride + pw.side_att;
        a.qbnd = pw.qkv_bnd; a.kgb = kb; a.vgb = kb + T;
        a.f16_single = X.attn_f16;
    }
    rc = launch_attention(a, st);
    if (rc) return rc;

    // ---- message = norm1(merge(o)) -> plane image
    reset();
    for_sides(xs, [&](int side) {
        PgProblem& p = add();
        p.A0 = at(pw.att_img, pw.side_att, side); p.bnd0 = pw.att_bnd + r0(side); p.nc0 = H * dp / 16;
        p.W = L.merge; p.nblk = 1; p.rows = nrows(side); p.C = C; p.mode = PG_LN; p.k_alg = C;
        p.gamma = W.norm1_w; p.beta = W.norm1_b; p.lnB = L.lnB1;
        p.pimg = at(pw.msg_img, pw.side_C, side); p.p_nct = nC; p.pbnd = pw.msg_bnd + r0(side);
        xk(p, (size_t)(p.rows + 63) / 64);
    });
    ++pw.xk_epoch;
    rc = launch_pgemm(g, st);
    if (rc) return rc;
    // ---- hidden = relu(mlp0([x | message])) -> plane image
    reset();
    for_sides(xs, [&](int side) {
        PgProblem& p = add();
        p.A0 = at(xin.img, pw.side_C, side); p.bnd0 = xin.bnd + r0(side); p.nc0 = nC;
        p.A1 = at(pw.msg_img, pw.side_C, side); p.bnd1 = pw.msg_bnd + r0(side); p.nc1 = nC;
        p.W = L.mlp0; p.nblk = 2; p.rows = nrows(side); p.C = C; p.mode = PG_PLANES; p.relu = 1; p.scale = 1.f;
        p.pimg = at(pw.hid_img, pw.side_hid, side); p.p_nct = 2 * nC; p.pbnd = pw.hid_bnd + r0(side);
        if (p.W.sub == 2) xk(p, xk_wide_units(p));
    });
    ++pw.xk_epoch;
    rc = launch_pgemm(g, st);
    if (rc) return rc;
    // ---- out = x + norm2(mlp2(hidden)) -> fp32 rows (the residual stream) + plane image (the next GEMMs' operand)
    reset();
    for_sides(xs, [&](int side) {
        PgProblem& p = add();
        p.A0 = at(pw.hid_img, pw.side_hid, side); p.bnd0 = pw.hid_bnd + r0(side); p.nc0 = 2 * nC;
        p.W = L.mlp2; p.nblk = 1; p.rows = nrows(side); p.C = C; p.mode = PG_LN;
        p.gamma = W.norm2_w; p.beta = W.norm2_b; p.lnB = L.lnB2;
        p.resid = xin.f32 + (size_t)r0(side) * C; p.ldr = C; p.bnd_res = xin.bnd + r0(side);
        p.out = out.f32 + (size_t)r0(side) * C; p.ldo = C;
        p.pimg = at(out.img, pw.side_C, side); p.p_nct = nC; p.pbnd = out.bnd + r0(side);
        xk(p, (size_t)(p.rows + 63) / 64);
    });
    ++pw.xk_epoch;
    return launch_pgemm(g, st);
}

// workspace of one denoiser + matching-head evaluation
struct DenoiseWs {
    LayerWs lw;
    float *fa, *fb, *cosT, *sinT, *proj, *sim;
    float *tgt_l0, *kv_l1;      // step-invariant: layer-0 output of the tgt rows, layer-1 K|V of those rows
    PlanesWs pl;
    const Prepack* pp;          // packed weights of the plane path (set per call)
    static void carve(Carver& c, DenoiseWs& w, const dr_loop_config& cfg, int P, int N, int M) {
        const int C = cfg.C;
        const size_t T = (size_t)P * (N + M);
        LayerWs::carve(c, w.lw, T, C);
        PlanesWs::carve(c, w.pl, cfg, P, N, M);
        w.tgt_l0 = c.take<float>(T * C);
        w.kv_l1 = c.take<float>(T * 2 * C);
        w.fa = c.take<float>(T * C);
        w.fb = c.take<float>(T * C);
        w.cosT = c.take<float>(T * (C / 2));
        w.sinT = c.take<float>(T * (C / 2));
        w.proj = c.take<float>(T * C);
        w.sim = c.take<float>((size_t)P * N * M);
        if (w.pl.on) { w.pl.fa.f32 = w.fa; w.pl.fb.f32 = w.fb; w.pl.tgt_l0.f32 = w.tgt_l0; }
    }
};

// six layers self, cross, ... (pipeline.py:142; transformero.py:170-186) starting from feat0,
// then the matching head's projection + N x M similarity (matching.py:173-207).  PE tables must be
// filled.  On return *final points at the buffer holding the refined features and ws.sim holds sim.
// The tgt cloud never moves and layer 0 is a self layer, so the tgt half of layer 0 and the K|V projections of
// layer 1's first cross call do not depend on the step (the reference recomputes them 20 times): fill_tgt_cache
// evaluates them once per loop, denoiser_and_sim(use_cache) reuses them.  Results are bit-identical.
static int fill_tgt_cache(const dr_loop_config& cfg, const dr_loop_weights& w, int P, int N, int M, const float* feat0,
                          const uint8_t* tokmask, DenoiseWs& ws, hipStream_t st) {
    const int C = cfg.C, H = cfg.H, PN = P * N, PM = P * M;
    const Family self_t{PN, M, PN, M};
    if (ws.pl.on) {
        const PlCtx X{ws.pp, &ws.pl, &ws.lw, C, H, P, N, M, ws.cosT, ws.sinT, tokmask, (cfg.flags & DR_LOOP_ATTN_F16) ? 1 : 0};
        int rc = layer_call_planes(X, w.layers[0], 0, ws.pl.feat0, SIDE_TGT, ws.pl.feat0, SIDE_TGT, ws.pl.tgt_l0, self_t, nullptr, st);
        if (rc || cfg.n_layers < 2) return rc;
        return layer_call_planes(X, w.layers[1], 1, ws.pl.tgt_l0, 0, ws.pl.tgt_l0, SIDE_TGT, ws.pl.tgt_l0, self_t, nullptr, st, nullptr, ws.kv_l1);
    }
    int rc = layer_call(w.layers[0], C, H, P, feat0, PN, PM, feat0, PN, PM, ws.cosT, ws.sinT, tokmask, self_t, nullptr, ws.lw,
                        ws.tgt_l0, st);
    if (rc || cfg.n_layers < 2) return rc;
    return layer_call(w.layers[1], C, H, P, nullptr, 0, 0, ws.tgt_l0, PN, PM, ws.cosT, ws.sinT, tokmask, self_t, nullptr, ws.lw,
                      nullptr, st, nullptr, ws.kv_l1);
}

static int denoiser_and_sim(const dr_loop_config& cfg, const dr_loop_weights& w, int P, int N, int M, const float* feat0,
                            const uint8_t* tokmask, DenoiseWs& ws, const float** final_feats, hipStream_t st,
                            bool use_cache = false) {
    const int C = cfg.C, H = cfg.H, T = P * (N + M), PN = P * N, PM = P * M;
    if (ws.pl.on) {
        const PlCtx X{ws.pp, &ws.pl, &ws.lw, C, H, P, N, M, ws.cosT, ws.sinT, tokmask, (cfg.flags & DR_LOOP_ATTN_F16) ? 1 : 0};
        const Family self_s{0, N, 0, N}, self_t{PN, M, PN, M}, cross_s{0, N, PN, M}, cross_t{PN, M, 0, N};
        const Tok* cur = &ws.pl.feat0;
        const Tok* bufs[2] = {&ws.pl.fa, &ws.pl.fb};
        int which = 0;
        for (int l = 0; l < cfg.n_layers; ++l) {
            const Tok* nxt = bufs[which];
            int rc;
            if (use_cache && l == 0) {
                // the tgt half of layer 0 is the cache (rows, plane image, bounds): the src half is written beside it, into the cache's
                // own token buffer -- layer 1 reads both halves there and nothing later writes to it (no copy of the cached half)
                nxt = &ws.pl.tgt_l0;
                rc = layer_call_planes(X, w.layers[0], 0, *cur, SIDE_SRC, *cur, SIDE_SRC, *nxt, self_s, nullptr, st);
                if (rc) return rc;
                cur = nxt;
                continue;
            } else if (l % 2 == 0) {
                rc = layer_call_planes(X, w.layers[l], l, *cur, SIDE_BOTH, *cur, SIDE_BOTH, *nxt, self_s, &self_t, st);
                if (rc) return rc;
            } else {
                // src attends tgt, then tgt attends the UPDATED src (quirk Q11)
                rc = layer_call_planes(X, w.layers[l], l, *cur, SIDE_SRC, *cur, SIDE_TGT, *nxt, cross_s, nullptr, st,
                                       (use_cache && l == 1) ? ws.kv_l1 : nullptr);
                if (rc) return rc;
                rc = layer_call_planes(X, w.layers[l], l, *cur, SIDE_TGT, *nxt, SIDE_SRC, *nxt, cross_t, nullptr, st);
                if (rc) return rc;
            }
            cur = nxt;
            which ^= 1;
        }
        *final_feats = cur->f32;
        if (env_knob("DR_HEAD_F32", 0)) {
            // (experiment: the head's projection on the f32-input MFMA GEMM, 24-bit operands, from the fp32 rows of the last layer)
            GemmBatch gf;
            memset(&gf, 0, sizeof(gf));
            GemmProblem& pf = gf.p[0];
            pf.A = cur->f32; pf.W = w.src_proj; pf.out = ws.proj; pf.rows = T; pf.ncols = C; pf.K = C; pf.K1 = C; pf.lda = C; pf.ldo = C;
            pf.epi = EPI_ROTARY; pf.rot_C = C; pf.cosT = ws.cosT; pf.sinT = ws.sinT; pf.scale = 1.0f / sqrtf((float)C);
            gf.n = 1;
            int rcf = launch_gemm(gf, st);
            if (rcf) return rcf;
            GemmBatch gs;
            memset(&gs, 0, sizeof(gs));
            GemmProblem& q = gs.p[0];
            q.A = ws.proj; q.W = ws.proj + (size_t)PN * C; q.out = ws.sim;
            q.rows = N; q.ncols = M; q.K = C; q.K1 = C; q.lda = C; q.ldo = M; q.epi = EPI_NONE; q.scale = 1.f;
            q.nbatch = P; q.sA = (long long)N * C; q.sW = (long long)M * C; q.sO = (long long)N * M;
            gs.n = 1;
            return launch_gemm(gs, st);
        }
        // matching head: src_proj on BOTH sides (quirk Q1), rotary, / sqrt(C)
        PgBatch g;
        memset(&g, 0, sizeof(g));
        size_t xk_next = 0;
        for (int side = 1; side <= 2; ++side) {
            PgProblem& p = g.p[g.n++];
            const int r0 = side == SIDE_TGT ? PN : 0;
            p.A0 = cur->img + (side == SIDE_TGT ? ws.pl.side_C : 0); p.bnd0 = cur->bnd + r0; p.nc0 = C / 16;
            p.W = ws.pp->head; p.nblk = 1; p.rows = side == SIDE_TGT ? PM : PN; p.C = C; p.mode = PG_F32;
            p.out = ws.proj + (size_t)r0 * C; p.ldo = C; p.blk_stride = 0; p.rot_mask = 1; p.rot_C = C; p.scale = 1.0f / sqrtf((float)C);
            p.cosT = ws.cosT + (size_t)r0 * (C / 2); p.sinT = ws.sinT + (size_t)r0 * (C / 2);
            p.csT = ws.pl.csT + (size_t)r0 * C;
            if (p.W.sub == 2) xk_assign(ws.pl, C, p, xk_next, xk_wide_units(p));
        }
        ++ws.pl.xk_epoch;
        int rc = launch_pgemm(g, st);
        if (rc) return rc;
        GemmBatch gs;
        memset(&gs, 0, sizeof(gs));
        GemmProblem& q = gs.p[0];
        q.A = ws.proj; q.W = ws.proj + (size_t)PN * C; q.out = ws.sim;
        q.rows = N; q.ncols = M; q.K = C; q.K1 = C; q.lda = C; q.ldo = M; q.epi = EPI_NONE; q.scale = 1.f;
        q.nbatch = P; q.sA = (long long)N * C; q.sW = (long long)M * C; q.sO = (long long)N * M;
        gs.n = 1;
        return launch_gemm(gs, st);
    }
    const float* cur = feat0;
    float* bufs[2] = {ws.fa, ws.fb};
    int which = 0;
    const Family self_s{0, N, 0, N}, self_t{PN, M, PN, M}, cross_s{0, N, PN, M}, cross_t{PN, M, 0, N};
    for (int l = 0; l < cfg.n_layers; ++l) {
        float* nxt = bufs[which];
        int rc;
        if (use_cache && l == 0) {
            // src half only, written beside the cached tgt half (into the cache's own buffer: nothing later writes to it)
            nxt = ws.tgt_l0;
            rc = layer_call(w.layers[0], C, H, P, cur, 0, PN, cur, 0, PN, ws.cosT, ws.sinT, tokmask, self_s, nullptr, ws.lw, nxt, st);
            if (rc) return rc;
            cur = nxt;
            continue;
        } else if (use_cache && l == 1) {
            rc = layer_call(w.layers[1], C, H, P, cur, 0, PN, cur, PN, PM, ws.cosT, ws.sinT, tokmask, cross_s, nullptr, ws.lw, nxt, st,
                            ws.kv_l1);
            if (rc) return rc;
            rc = layer_call(w.layers[1], C, H, P, cur, PN, PM, nxt, 0, PN, ws.cosT, ws.sinT, tokmask, cross_t, nullptr, ws.lw, nxt, st);
            if (rc) return rc;
        } else if (l % 2 == 0) {
            rc = layer_call(w.layers[l], C, H, P, cur, 0, T, cur, 0, T, ws.cosT, ws.sinT, tokmask, self_s, &self_t, ws.lw, nxt, st);
            if (rc) return rc;
        } else {
            // src attends tgt, then tgt attends the UPDATED src (quirk Q11)
            rc = layer_call(w.layers[l], C, H, P, cur, 0, PN, cur, PN, PM, ws.cosT, ws.sinT, tokmask, cross_s, nullptr, ws.lw, nxt, st);
            if (rc) return rc;
            rc = layer_call(w.layers[l], C, H, P, cur, PN, PM, nxt, 0, PN, ws.cosT, ws.sinT, tokmask, cross_t, nullptr, ws.lw, nxt, st);
            if (rc) return rc;
        }
        cur = nxt;
        which ^= 1;
    }
    *final_feats = cur;
    // matching head: src_proj on BOTH sides (quirk Q1), rotary, / sqrt(C)
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    GemmProblem& p = g.p[0];
    p.A = cur; p.W = w.src_proj; p.out = ws.proj; p.rows = T; p.ncols = C; p.K = C; p.K1 = C; p.lda = C; p.ldo = C;
    p.epi = EPI_ROTARY; p.rot_C = C; p.cosT = ws.cosT; p.sinT = ws.sinT; p.scale = 1.0f / sqrtf((float)C);
    g.n = 1;
    int rc = launch_gemm(g, st);
    if (rc) return rc;
    // sim[p] = a_p b_p^T : one NT GEMM per pair, all P pairs as one strided batch
    memset(&g, 0, sizeof(g));
    GemmProblem& q = g.p[0];
    q.A = ws.proj; q.W = ws.proj + (size_t)PN * C; q.out = ws.sim;
    q.rows = N; q.ncols = M; q.K = C; q.K1 = C; q.lda = C; q.ldo = M; q.epi = EPI_NONE; q.scale = 1.f;
    q.nbatch = P; q.sA = (long long)N * C; q.sW = (long long)M * C; q.sO = (long long)N * M;
    g.n = 1;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    return DR_OK;
}

static int fill_pe(const dr_loop_config& cfg, const dr_loop_weights& w, int P, int N, int M, const float* s_pcd,
                   const float* Rf, const float* tf, const float* t_pcd, bool do_src, bool do_tgt, DenoiseWs& ws,
                   hipStream_t st) {
    const int halfC = cfg.C / 2;
    int rc = DR_OK;
    if (do_src)
        rc = launch_vol_pe(s_pcd, P * N, N, Rf, tf, cfg.C, cfg.origin[0], cfg.origin[1], cfg.origin[2], cfg.voxel, w.pe_freq,
                           ws.cosT, ws.sinT, st, ws.pl.on ? ws.pl.csT : nullptr);
    if (rc == DR_OK && do_tgt)
        rc = launch_vol_pe(t_pcd, P * M, M, nullptr, nullptr, cfg.C, cfg.origin[0], cfg.origin[1], cfg.origin[2], cfg.voxel,
                           w.pe_freq, ws.cosT + (size_t)P * N * halfC, ws.sinT + (size_t)P * N * halfC, st,
                           ws.pl.on ? ws.pl.csT + (size_t)P * N * halfC * 2 : nullptr);
    return rc;
}

struct LoopWs {
    DenoiseWs dw;
    float *feat0, *wconf, *x0, *R, *t, *Rf, *tf, *conf32;
    double *x, *dmin, *cond;
    void* pmin;                              // pair_min_scratch_bytes(P): slice minima + arrival counters of the multi-workgroup minimum
    int* ok;
    uint8_t* tokmask;
    void* skws;
    size_t skws_bytes;
    void* pws;
    size_t pws_bytes;
    unsigned* status;                        // the call's own sticky status word (dr_denoise_loop_status): zeroed when a call starts
    static size_t carve(Carver& c, LoopWs& w, const dr_loop_config& cfg, int P, int N, int M) {
        const size_t T = (size_t)P * (N + M), NM = (size_t)P * N * M;
        w.status = c.take<unsigned>(4);      // (first: its place does not depend on the configuration)
        DenoiseWs::carve(c, w.dw, cfg, P, N, M);
        w.feat0 = c.take<float>(T * cfg.C);
        if (w.dw.pl.on) w.dw.pl.feat0.f32 = w.feat0;
        w.wconf = c.take<float>(NM);
        w.x0 = c.take<float>(NM);
        w.conf32 = c.take<float>(NM);
        w.x = c.take<double>(NM);
        w.dmin = c.take<double>(P);
        w.cond = c.take<double>(P);
        w.pmin = (void*)c.take<char>(pair_min_scratch_bytes(P));
        w.R = c.take<float>((size_t)P * 9);
        w.t = c.take<float>((size_t)P * 3);
        w.Rf = c.take<float>((size_t)P * 9);
        w.tf = c.take<float>((size_t)P * 3);
        w.ok = c.take<int>(P);
        w.tokmask = c.take<uint8_t>(T);
        const int strict = (cfg.flags & DR_LOOP_STRICT_F64) ? DR_SK_STRICT : 0;
        size_t a = dr_sinkhorn_workspace_bytes(P, N, M, 8, strict);
        size_t b = dr_sinkhorn_workspace_bytes(P, N, M, 4, 0);
        w.skws_bytes = a > b ? a : b;
        w.skws = w.skws_bytes ? (void*)c.take<char>(w.skws_bytes) : nullptr;
        w.pws_bytes = procrustes_workspace_bytes(P, N, M);
        w.pws = w.pws_bytes ? (void*)c.take<char>(w.pws_bytes) : nullptr;
        return c.off + 256;
    }
};

// plane path, once per call: the packed weights (the caller's, or packed now into the workspace) and the plane image of the
// external features with their row maxima as bounds
static int planes_begin(const dr_loop_config& cfg, const dr_loop_weights& w, int P, int N, int M, DenoiseWs& ws, Prepack& pp,
                        hipStream_t st) {
    if (!ws.pl.on) return DR_OK;
    void* buf = const_cast<void*>(w.prepacked);
    if (!buf) {
        buf = ws.pl.own_pack;
        const int rc = Prepack::fill(buf, cfg, w, st);
        if (rc) return rc;
    }
    Prepack::carve(buf, cfg, &pp);
    ws.pp = &pp;
    const int C = cfg.C, PN = P * N, PM = P * M;
    if (ws.pl.xk_flags) {
        DR_HIP_CHECK(hipMemsetAsync(ws.pl.xk_flags, 0, pgemm_xk_flag_bytes(), st));
        ws.pl.xk_epoch = 0;
    }
    int rc = launch_planes_from_f32(ws.pl.feat0.f32, C, PN, C, ws.pl.feat0.img, ws.pl.feat0.bnd, st);
    if (rc == DR_OK) rc = launch_planes_from_f32(ws.pl.feat0.f32 + (size_t)PN * C, C, PM, C, ws.pl.feat0.img + ws.pl.side_C, ws.pl.feat0.bnd + PN, st);
    return rc;
}

}  // namespace dr

using namespace dr;

extern "C" {

int dr_init(void) {
    int rc = attention_configure();
    if (rc == DR_OK) rc = gemm_configure();
    if (rc == DR_OK) rc = pgemm_configure();
    return rc;
}

/* diagnostics for tools/: force the GEMM tile configuration (-1 auto, 0 small, 1 medium, 2 large) */
void dr_debug_enable_env(int on) { enable_env_knobs(on != 0); }
void dr_debug_gemm_config(int c) { gemm_force_config(c); }
void dr_debug_attention_config(int flash_min_workgroups) { attention_force_flash_min(flash_min_workgroups); }
void dr_debug_attention_split(int on) { attention_force_split(on); }

int dr_vol_pe_f32(int rows, int rows_per_pair, int C, const float* xyz, const float* R, const float* t, float origin_x,
                  float origin_y, float origin_z, float voxel, const float* freq, float* cos_out, float* sin_out,
                  void* stream) {
    if (rows < 0 || C <= 0 || !xyz || !freq || !cos_out || !sin_out || rows_per_pair < 1) return DR_EINVAL;
    if ((R == nullptr) != (t == nullptr)) return DR_EINVAL;
    return launch_vol_pe(xyz, rows, rows_per_pair, R, t, C, origin_x, origin_y, origin_z, voxel, freq, cos_out, sin_out,
                         (hipStream_t)stream);
}

int dr_linear_f32(int rows, int ncols, int K, const float* x, const float* W, float* out, int epilogue, const float* cos_t,
                  const float* sin_t, int rot_C, float scale, void* stream) {
    if (rows < 0 || ncols <= 0 || K <= 0 || !x || !W || !out) return DR_EINVAL;
    if ((epilogue & EPI_ROTARY) && (!cos_t || !sin_t || rot_C <= 0 || (rot_C & 1))) return DR_EINVAL;
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    GemmProblem& p = g.p[0];
    p.A = x; p.W = W; p.out = out; p.rows = rows; p.ncols = ncols; p.K = K; p.K1 = K; p.lda = K; p.ldo = ncols;
    p.epi = epilogue; p.cosT = cos_t; p.sinT = sin_t; p.rot_C = rot_C; p.scale = scale;
    g.n = 1;
    return launch_gemm(g, (hipStream_t)stream);
}

int dr_gemm_nt_batched_f32(int nbatch, int rows, int ncols, int K, const float* A, long long stride_a, const float* W, long long stride_w, float* out,
                           long long stride_o, float scale, void* stream) {
    if (nbatch < 0 || rows < 0 || ncols <= 0 || K <= 0 || (K & 3) || !A || !W || !out) return DR_EINVAL;
    if (nbatch == 0 || rows == 0) return DR_OK;
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    GemmProblem& p = g.p[0];
    p.A = A; p.W = W; p.out = out; p.rows = rows; p.ncols = ncols; p.K = K; p.K1 = K; p.lda = K; p.ldo = ncols; p.epi = EPI_NONE; p.scale = scale;
    p.nbatch = nbatch; p.sA = stride_a; p.sW = stride_w; p.sO = stride_o;
    g.n = 1;
    return launch_gemm(g, (hipStream_t)stream);
}

int dr_linear_ex_f32(int rows, int ncols, int K, const float* x, int lda, const float* W, const float* bias, float* out, int ldo,
                     int epilogue, float scale, void* stream) {
    if (rows < 0 || ncols <= 0 || K <= 0 || !x || !W || !out || lda < K || ldo < ncols || (epilogue & EPI_ROTARY)) return DR_EINVAL;
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    GemmProblem& p = g.p[0];
    p.A = x; p.W = W; p.bias = bias; p.out = out; p.rows = rows; p.ncols = ncols; p.K = K; p.K1 = K; p.lda = lda; p.ldo = ldo;
    p.epi = epilogue; p.scale = scale;
    g.n = 1;
    return launch_gemm(g, (hipStream_t)stream);
}

size_t dr_attention_layer_workspace_bytes(int P, int Lx, int Ly, int C) {
    Carver c(nullptr, 0);
    LayerWs w;
    const size_t T = (size_t)P * (Lx + Ly);
    LayerWs::carve(c, w, T, C);
    c.take<float>(T * C);          // x|y token buffer
    c.take<float>(T * C);          // output token buffer
    c.take<float>(T * (C / 2));    // cos
    c.take<float>(T * (C / 2));    // sin
    c.take<uint8_t>(T);
    c.take<float>(T * C);          // q | k inputs of the additive (sinusoidal) form
    return c.off + 256;
}

int dr_attention_layer_f32(const dr_layer_weights* w, int C, int H, int P, int Lx, int Ly, const float* x, const float* y,
                           const float* cos_x, const float* sin_x, const float* cos_y, const float* sin_y,
                           const uint8_t* x_mask, const uint8_t* y_mask, float* out, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!cos_x || !sin_x || !cos_y || !sin_y) return DR_EINVAL;
    return dr_attention_layer_pe_f32(w, C, H, P, Lx, Ly, x, y, nullptr, nullptr, cos_x, sin_x, cos_y, sin_y, x_mask, y_mask, out, workspace,
                                     workspace_bytes, stream);
}

int dr_attention_layer_pe_f32(const dr_layer_weights* w, int C, int H, int P, int Lx, int Ly, const float* x, const float* y,
                              const float* xq, const float* yk, const float* cos_x, const float* sin_x, const float* cos_y, const float* sin_y,
                              const uint8_t* x_mask, const uint8_t* y_mask, float* out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!w || !x || !y || !out || P < 1 || Lx < 1 || Ly < 1 || C % H || (C / H) % 4 || C % 4) return DR_EINVAL;
    if ((x_mask == nullptr) != (y_mask == nullptr)) return DR_EINVAL;
    const bool rotary = cos_x != nullptr;
    if ((sin_x != nullptr) != rotary || (cos_y != nullptr) != rotary || (sin_y != nullptr) != rotary) return DR_EINVAL;
    if ((xq == nullptr) != (yk == nullptr) || (xq && rotary)) return DR_EINVAL;   // additive code and rotary code exclude each other (pe_type)
    if (workspace_bytes < dr_attention_layer_workspace_bytes(P, Lx, Ly, C) || !workspace) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    LayerWs lw;
    const size_t T = (size_t)P * (Lx + Ly), PX = (size_t)P * Lx, PY = (size_t)P * Ly;
    const int halfC = C / 2;
    LayerWs::carve(c, lw, T, C);
    float* tok = c.take<float>(T * C);
    float* otok = c.take<float>(T * C);
    float* cosT = c.take<float>(T * halfC);
    float* sinT = c.take<float>(T * halfC);
    uint8_t* mask = c.take<uint8_t>(T);
    float* qk_buf = c.take<float>(T * C);
    DR_HIP_CHECK(hipMemcpyAsync(tok, x, PX * C * 4, hipMemcpyDeviceToDevice, st));
    DR_HIP_CHECK(hipMemcpyAsync(tok + PX * C, y, PY * C * 4, hipMemcpyDeviceToDevice, st));
    float* qk_tok = nullptr;                       // q | k inputs of the additive form
    if (rotary) {
        DR_HIP_CHECK(hipMemcpyAsync(cosT, cos_x, PX * halfC * 4, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(cosT + PX * halfC, cos_y, PY * halfC * 4, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(sinT, sin_x, PX * halfC * 4, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(sinT + PX * halfC, sin_y, PY * halfC * 4, hipMemcpyDeviceToDevice, st));
    } else if (xq) {
        qk_tok = qk_buf;
        DR_HIP_CHECK(hipMemcpyAsync(qk_tok, xq, PX * C * 4, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(qk_tok + PX * C, yk, PY * C * 4, hipMemcpyDeviceToDevice, st));
    }
    if (x_mask) {
        DR_HIP_CHECK(hipMemcpyAsync(mask, x_mask, PX, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(mask + PX, y_mask, PY, hipMemcpyDeviceToDevice, st));
    }
    const Family f{0, Lx, (int)PX, Ly};
    int rc = layer_call(*w, C, H, P, tok, 0, (int)PX, tok, (int)PX, (int)PY, rotary ? cosT : nullptr, rotary ? sinT : nullptr,
                        x_mask ? mask : nullptr, f, nullptr, lw, otok, st, nullptr, nullptr, qk_tok, qk_tok);
    if (rc) return rc;
    DR_HIP_CHECK(hipMemcpyAsync(out, otok, PX * C * 4, hipMemcpyDeviceToDevice, st));
    return DR_OK;
}

size_t dr_procrustes_workspace_bytes(int P, int N, int M) {
    return (P < 1 || N < 1 || M < 1) ? 0 : procrustes_workspace_bytes(P, N, M);
}

int dr_procrustes_f32(int P, int N, int M, const float* conf, const float* src_pcd, const float* tgt_pcd,
                      const uint8_t* src_mask, const uint8_t* tgt_mask, int use_mask_len, float sample_rate,
                      float max_condition_num, float* R, float* t, float* R_forwd, float* t_forwd, double* condition,
                      int32_t* solution_mask, int32_t* topk_idx, void* workspace, size_t workspace_bytes, void* stream) {
    if (P < 0 || N < 1 || M < 1 || !conf || !src_pcd || !tgt_pcd || !R || !t || !R_forwd || !t_forwd || !condition || !solution_mask)
        return DR_EINVAL;
    if (P == 0) return DR_OK;
    // tiles beyond 256 x 256 select with the whole chip: the caller's scratch (header contract: the caller owns every buffer)
    const size_t wsb = procrustes_workspace_bytes(P, N, M);
    if (wsb && (!workspace || workspace_bytes < wsb)) return DR_EWORKSPACE;
    return launch_procrustes(conf, src_pcd, tgt_pcd, src_mask, tgt_mask, P, N, M, use_mask_len, sample_rate, max_condition_num,
                             R, t, R_forwd, t_forwd, condition, solution_mask, topk_idx, (hipStream_t)stream, wsb ? workspace : nullptr, wsb);
}

size_t dr_top1_union_workspace_bytes(int P, int N, int M, int elem_bytes) {
    return (P < 1 || N < 1 || M < 1 || (elem_bytes != 4 && elem_bytes != 8)) ? 0 : top1_union_workspace_bytes(P, N, M, (size_t)elem_bytes);
}

}  // extern "C"
template <typename T>
static int top1_union_entry(const T* conf, int P, int N, int M, int64_t* matches, int32_t* count, void* ws, size_t ws_bytes, hipStream_t st) {
    if (P < 0 || N < 1 || M < 1 || !conf || !matches || !count) return DR_EINVAL;
    if (P == 0) return DR_OK;
    const size_t wsb = top1_union_workspace_bytes(P, N, M, sizeof(T));
    if (wsb && (!ws || ws_bytes < wsb)) return DR_EWORKSPACE;
    return launch_top1_union<T>(conf, P, N, M, (long long*)matches, count, st, nullptr, nullptr, wsb ? ws : nullptr, wsb);
}
extern "C" {
int dr_top1_union_f64(int P, int N, int M, const double* conf, int64_t* matches, int32_t* count, void* workspace, size_t workspace_bytes,
                      void* stream) {
    return top1_union_entry<double>(conf, P, N, M, matches, count, workspace, workspace_bytes, (hipStream_t)stream);
}
int dr_top1_union_f32(int P, int N, int M, const float* conf, int64_t* matches, int32_t* count, void* workspace, size_t workspace_bytes,
                      void* stream) {
    return top1_union_entry<float>(conf, P, N, M, matches, count, workspace, workspace_bytes, (hipStream_t)stream);
}

static int check_cfg(const dr_loop_config* cfg, const dr_loop_weights* w, int P, int N, int M) {
    if (!cfg || !w || P < 1 || N < 1 || M < 1) return DR_EINVAL;
    if (cfg->C % cfg->H || (cfg->C / cfg->H) % 4 || cfg->C % 6 || cfg->n_layers < 1 || cfg->sk_iters < 1) return DR_EINVAL;
    if (!w->layers || !w->src_proj || !w->bin_score || !w->pe_freq) return DR_EINVAL;
    if (cfg->variant != DR_VARIANT_3DMATCH && cfg->variant != DR_VARIANT_4DMATCH) return DR_EINVAL;
    return DR_OK;
}

size_t dr_loop_prepack_bytes(const dr_loop_config* cfg) {
    if (!cfg || !Prepack::supported(*cfg)) return 0;
    return Prepack::carve(nullptr, *cfg, nullptr);
}

int dr_loop_prepack(const dr_loop_config* cfg, const dr_loop_weights* w, void* packed, size_t packed_bytes, void* stream) {
    if (!cfg || !w || !w->layers || !w->src_proj || !packed || ((uintptr_t)packed & 255)) return DR_EINVAL;
    if (!Prepack::supported(*cfg)) return DR_ENOSUP;
    if (packed_bytes < Prepack::carve(nullptr, *cfg, nullptr)) return DR_EWORKSPACE;
    return Prepack::fill(packed, *cfg, *w, (hipStream_t)stream);
}

size_t dr_denoise_loop_workspace_bytes(const dr_loop_config* cfg, int P, int N, int M) {
    if (!cfg || P < 1 || N < 1 || M < 1) return 0;
    Carver c(nullptr, 0);
    LoopWs w;
    return LoopWs::carve(c, w, *cfg, P, N, M);
}

int dr_denoise_loop_status(void* workspace, void* stream, int clear) {
    if (!workspace) return DR_EINVAL;
    Carver c(workspace, (size_t)-1);
    return sinkhorn_call_status(c.take<unsigned>(4), (hipStream_t)stream, clear != 0);
}

int dr_denoiser_match_f32(const dr_loop_config* cfg, const dr_loop_weights* w, int P, int N, int M, const float* src_feats,
                          const float* tgt_feats, const float* s_pcd_warped, const float* t_pcd, const uint8_t* src_mask,
                          const uint8_t* tgt_mask, float* src_out, float* tgt_out, float* conf, void* workspace,
                          size_t workspace_bytes, void* stream) {
    int rc = check_cfg(cfg, w, P, N, M);
    if (rc) return rc;
    if (!src_feats || !tgt_feats || !s_pcd_warped || !t_pcd || !conf) return DR_EINVAL;
    if ((src_mask == nullptr) != (tgt_mask == nullptr)) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_denoise_loop_workspace_bytes(cfg, P, N, M)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    LoopWs L;
    LoopWs::carve(c, L, *cfg, P, N, M);
    DR_HIP_CHECK(hipMemsetAsync(L.status, 0, 16, st));          // the status of THIS call (one 16-byte fill per call)
    L.dw.pl.status = L.status;
    const int C = cfg->C;
    const size_t PN = (size_t)P * N, PM = (size_t)P * M;
    DR_HIP_CHECK(hipMemcpyAsync(L.feat0, src_feats, PN * C * 4, hipMemcpyDeviceToDevice, st));
    DR_HIP_CHECK(hipMemcpyAsync(L.feat0 + PN * C, tgt_feats, PM * C * 4, hipMemcpyDeviceToDevice, st));
    if (src_mask) {
        DR_HIP_CHECK(hipMemcpyAsync(L.tokmask, src_mask, PN, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(L.tokmask + PN, tgt_mask, PM, hipMemcpyDeviceToDevice, st));
    }
    Prepack pp;
    rc = planes_begin(*cfg, *w, P, N, M, L.dw, pp, st);
    if (rc) return rc;
    rc = fill_pe(*cfg, *w, P, N, M, s_pcd_warped, nullptr, nullptr, t_pcd, true, true, L.dw, st);
    if (rc) return rc;
    const float* fin = nullptr;
    rc = denoiser_and_sim(*cfg, *w, P, N, M, L.feat0, src_mask ? L.tokmask : nullptr, L.dw, &fin, st);
    if (rc) return rc;
    if (src_out) DR_HIP_CHECK(hipMemcpyAsync(src_out, fin, PN * C * 4, hipMemcpyDeviceToDevice, st));
    if (tgt_out) DR_HIP_CHECK(hipMemcpyAsync(tgt_out, fin + PN * C, PM * C * 4, hipMemcpyDeviceToDevice, st));
    return sinkhorn_f32(P, N, M, L.dw.sim, src_mask, tgt_mask, w->bin_score, cfg->sk_iters,
                        DR_SK_OUT_CONF | (src_mask ? DR_SK_APPLY_MASK : 0), conf, L.skws, L.skws_bytes, st, L.status);
}

int dr_denoise_loop(const dr_loop_config* cfg, const dr_loop_weights* w, int P, int N, int M, const float* src_feats,
                    const float* tgt_feats, const float* s_pcd, const float* t_pcd, const uint8_t* src_mask,
                    const uint8_t* tgt_mask, const float* x_T, const float* noise, double* conf, double* x_final,
                    int64_t* matches, int32_t* match_count, float* R_final, float* t_final, const dr_loop_trace* trace,
                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_cfg(cfg, w, P, N, M);
    if (rc) return rc;
    if (!src_feats || !tgt_feats || !s_pcd || !t_pcd || !x_T || !conf || cfg->steps < 1 || !cfg->h_alphas_cumprod || !cfg->h_times)
        return DR_EINVAL;
    if ((src_mask == nullptr) != (tgt_mask == nullptr)) return DR_EINVAL;
    const bool v4d = cfg->variant == DR_VARIANT_4DMATCH;
    if (v4d && !noise) return DR_EINVAL;
    if ((matches == nullptr) != (match_count == nullptr) || (R_final == nullptr) != (t_final == nullptr)) return DR_EINVAL;
    if (trace && (trace->force_R == nullptr) != (trace->force_t == nullptr)) return DR_EINVAL;
    if (!workspace || workspace_bytes < dr_denoise_loop_workspace_bytes(cfg, P, N, M)) return DR_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    LoopWs L;
    LoopWs::carve(c, L, *cfg, P, N, M);
    DR_HIP_CHECK(hipMemsetAsync(L.status, 0, 16, st));          // the status of THIS call (one 16-byte fill per call)
    L.dw.pl.status = L.status;
    const int C = cfg->C;
    const size_t PN = (size_t)P * N, PM = (size_t)P * M, NM = (size_t)P * N * M;
    const uint8_t* tokmask = src_mask ? L.tokmask : nullptr;
    const int strict = (cfg->flags & DR_LOOP_STRICT_F64) ? DR_SK_STRICT : 0;
    // DR_LOOP_RAGGED: the masks are the true extents of pairs padded to (N, M); every pair gets its unpadded result
    const bool ragged = (cfg->flags & DR_LOOP_RAGGED) && src_mask;
    const int mflag = src_mask ? (DR_SK_APPLY_MASK | (ragged ? DR_SK_RAGGED : 0)) : 0;
    const uint8_t* rsm = ragged ? src_mask : nullptr;
    const uint8_t* rtm = ragged ? tgt_mask : nullptr;

    DR_HIP_CHECK(hipMemcpyAsync(L.feat0, src_feats, PN * C * 4, hipMemcpyDeviceToDevice, st));
    DR_HIP_CHECK(hipMemcpyAsync(L.feat0 + PN * C, tgt_feats, PM * C * 4, hipMemcpyDeviceToDevice, st));
    if (src_mask) {
        DR_HIP_CHECK(hipMemcpyAsync(L.tokmask, src_mask, PN, hipMemcpyDeviceToDevice, st));
        DR_HIP_CHECK(hipMemcpyAsync(L.tokmask + PN, tgt_mask, PM, hipMemcpyDeviceToDevice, st));
    }
    rc = launch_f32_to_f64(x_T, L.x, NM, st);     // exact widening; step 1 keeps float32 semantics
    if (rc) return rc;
    if (!v4d && P <= 32)                          // arrival counters of the multi-workgroup minimum (few pairs: stateops.hip)
        DR_HIP_CHECK(hipMemsetAsync((char*)L.pmin + pair_min_scratch_bytes(P) - 64 * (size_t)P, 0, 64 * (size_t)P, st));
    Prepack pp;
    rc = planes_begin(*cfg, *w, P, N, M, L.dw, pp, st);
    if (rc) return rc;
    // the target cloud never moves: its position code is computed once (the reference recomputes it
    // every step, transformero.py:166)
    rc = fill_pe(*cfg, *w, P, N, M, s_pcd, nullptr, nullptr, t_pcd, false, true, L.dw, st);
    if (rc) return rc;
    rc = fill_tgt_cache(*cfg, *w, P, N, M, L.feat0, tokmask, L.dw, st);
    if (rc) return rc;

    const double* ac = cfg->h_alphas_cumprod;
    const float* fin = nullptr;
    for (int k = 0; k < cfg->steps; ++k) {
        const int tcur = cfg->h_times[k], tnext = cfg->h_times[k + 1];
        // teacher forcing (parity tests): this step starts from the caller's state, not from the loop's own
        if (trace && trace->force_x) DR_HIP_CHECK(hipMemcpyAsync(L.x, trace->force_x + (size_t)k * NM, NM * 8, hipMemcpyDeviceToDevice, st));
        // -- x <- x - x.min() (3D only, pipeline.py:239); mask; Sinkhorn; exp; slice; float32 (pipeline.py:293-302)
        const double* shift = nullptr;
        if (!v4d) {
            rc = launch_pair_min(L.x, P, N * M, L.dmin, st, M, rsm, rtm, P <= 32 ? L.pmin : nullptr);
            if (rc) return rc;
            shift = L.dmin;
        }
        rc = sinkhorn_f64(P, N, M, L.x, shift, src_mask, tgt_mask, w->bin_score, cfg->sk_iters,
                          DR_SK_OUT_CONF | DR_SK_OUT_F32 | mflag | (k > 0 ? strict : 0), L.wconf, L.skws, L.skws_bytes, st, L.status);
        if (rc) return rc;
        // -- denoising_soft_procrustes (pipeline.py:304)
        int* tk = nullptr;
        if (trace && trace->topk_idx) {
            const size_t Kf = (size_t)(int)((float)(N > M ? N : M) * cfg->sample_rate);
            tk = trace->topk_idx + (size_t)k * P * Kf;
            DR_HIP_CHECK(hipMemsetAsync(tk, 0xff, (size_t)P * Kf * 4, st));
        }
        if (trace && trace->wconf) DR_HIP_CHECK(hipMemcpyAsync(trace->wconf + (size_t)k * NM, L.wconf, NM * 4, hipMemcpyDeviceToDevice, st));
        rc = launch_procrustes(L.wconf, s_pcd, t_pcd, src_mask, tgt_mask, P, N, M, (v4d || ragged) ? 1 : 0, cfg->sample_rate,
                               cfg->max_condition_num, L.R, L.t, L.Rf, L.tf, L.cond, L.ok, tk, st, L.pws, L.pws_bytes);
        if (rc) return rc;
        if (trace && trace->R_forwd) DR_HIP_CHECK(hipMemcpyAsync(trace->R_forwd + (size_t)k * P * 9, L.Rf, (size_t)P * 36, hipMemcpyDeviceToDevice, st));
        if (trace && trace->t_forwd) DR_HIP_CHECK(hipMemcpyAsync(trace->t_forwd + (size_t)k * P * 3, L.tf, (size_t)P * 12, hipMemcpyDeviceToDevice, st));
        if (trace && trace->cond) DR_HIP_CHECK(hipMemcpyAsync(trace->cond + (size_t)k * P, L.cond, (size_t)P * 8, hipMemcpyDeviceToDevice, st));
        if (trace && trace->force_R) {           // teacher forcing: warp with the caller's pose (the fit above is traced all the same)
            DR_HIP_CHECK(hipMemcpyAsync(L.Rf, trace->force_R + (size_t)k * P * 9, (size_t)P * 36, hipMemcpyDeviceToDevice, st));
            DR_HIP_CHECK(hipMemcpyAsync(L.tf, trace->force_t + (size_t)k * P * 3, (size_t)P * 12, hipMemcpyDeviceToDevice, st));
        }
        // -- position code of the warped source (pipeline.py:306, transformero.py:165)
        rc = fill_pe(*cfg, *w, P, N, M, s_pcd, L.Rf, L.tf, t_pcd, true, false, L.dw, st);
        if (rc) return rc;
        // -- denoising_transformer + denoising_coarse_matching (pipeline.py:243-244)
        rc = denoiser_and_sim(*cfg, *w, P, N, M, L.feat0, tokmask, L.dw, &fin, st, true);
        if (rc) return rc;
        rc = sinkhorn_f32(P, N, M, L.dw.sim, src_mask, tgt_mask, w->bin_score, cfg->sk_iters, DR_SK_OUT_CONF | mflag, L.x0,
                          L.skws, L.skws_bytes, st, L.status);
        if (rc) return rc;
        if (trace && trace->x0) DR_HIP_CHECK(hipMemcpyAsync(trace->x0 + (size_t)k * NM, L.x0, NM * 4, hipMemcpyDeviceToDevice, st));
        // -- DDIM update (pipeline.py:246-256)
        const double a = ac[tcur], an = ac[tnext];
        DdimArgs d;
        d.x = L.x; d.x0 = L.x0; d.shift = shift; d.noise = v4d ? noise + (size_t)k * NM : nullptr;
        d.src_mask = src_mask; d.tgt_mask = tgt_mask; d.N = N; d.M = M; d.first_step = (k == 0);
        d.sra = sqrt(1.0 / a); d.srm1 = sqrt(1.0 / a - 1.0);
        d.sigma = 1.0 * sqrt((1.0 - a / an) * (1.0 - an) / (1.0 - a));
        d.c = sqrt(1.0 - an - d.sigma * d.sigma);
        d.sqrt_an = (float)sqrt(an);
        rc = launch_ddim(d, P, st);
        if (rc) return rc;
        if (trace && trace->x_next) DR_HIP_CHECK(hipMemcpyAsync(trace->x_next + (size_t)k * NM, L.x, NM * 8, hipMemcpyDeviceToDevice, st));
    }
    if (x_final) DR_HIP_CHECK(hipMemcpyAsync(x_final, L.x, NM * 8, hipMemcpyDeviceToDevice, st));
    if (trace && (trace->feats_nopos || trace->feats_pos) && fin) {
        // data["src_feats_nopos"] / ["src_feats"] (+ tgt) of the last Matching.forward (matching.py:177-187): src_proj on both
        // sides (quirk Q1), without and with the rotary embedding of the last step's position code
        for (int pos = 0; pos < 2; ++pos) {
            float* dst = pos ? trace->feats_pos : trace->feats_nopos;
            if (!dst) continue;
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            GemmProblem& p = g.p[0];
            p.A = fin; p.W = w->src_proj; p.out = dst; p.rows = (int)(PN + PM); p.ncols = C; p.K = C; p.K1 = C; p.lda = C; p.ldo = C;
            p.epi = pos ? EPI_ROTARY : EPI_NONE; p.rot_C = C; p.cosT = L.dw.cosT; p.sinT = L.dw.sinT; p.scale = 1.f;
            g.n = 1;
            rc = launch_gemm(g, st);
            if (rc) return rc;
        }
    }

    // -- read-out
    if (v4d) {
        rc = launch_sigmoid(L.x, conf, NM, st);                       // 4D/models/pipeline.py:192
        if (rc) return rc;
    } else {
        rc = launch_pair_min(L.x, P, N * M, L.dmin, st, M, rsm, rtm, P <= 32 ? L.pmin : nullptr);  // pipeline.py:264-272
        if (rc) return rc;
        rc = sinkhorn_f64(P, N, M, L.x, L.dmin, src_mask, tgt_mask, w->bin_score, cfg->sk_iters, DR_SK_OUT_CONF | mflag | strict,
                          conf, L.skws, L.skws_bytes, st, L.status);
        if (rc) return rc;
        if (matches) {
            // (the steps' x0 tile is free by now; the row-block arg-maxima need < N M floats)
            rc = launch_top1_union<double>(conf, P, N, M, (long long*)matches, match_count, st, rsm, rtm, L.x0, NM * 4);
            if (rc) return rc;
        }
    }
    if (R_final) {
        // soft_procrustes on float32(conf): the well-defined value of pipeline.py:282 (quirk Q3)
        rc = launch_f64_to_f32(conf, L.conf32, NM, st);
        if (rc) return rc;
        rc = launch_procrustes(L.conf32, s_pcd, t_pcd, src_mask, tgt_mask, P, N, M, (v4d || ragged) ? 1 : 0, cfg->sample_rate,
                               cfg->max_condition_num, R_final, t_final, L.Rf, L.tf, L.cond, L.ok, nullptr, st, L.pws, L.pws_bytes);
        if (rc) return rc;
    }
    return DR_OK;
}

}  // extern "C"
