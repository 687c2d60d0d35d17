// kernels.h -- internal launch interface between the .hip translation units of libdiffreg_hip.
#pragma once
#include "common.h"

namespace dr {

// ---------------------------------------------------------------------------------------------
// optional per-kernel-family timing with HIP events on the launch stream (dr_prof_* in the C ABI);
// costs one branch per launch when disabled.  Never enable inside a stream capture.
// ---------------------------------------------------------------------------------------------
enum ProfKind { PK_GEMM = 0, PK_ATTN, PK_LN, PK_PE, PK_SINKHORN, PK_PROCRUSTES, PK_STATE, PK_GEMM_SPLIT, PK_COUNT };
extern bool g_prof_on;
void prof_begin(int kind, double work, hipStream_t st);
void prof_end(int kind, hipStream_t st);
struct ProfScope {
    int kind; hipStream_t st; bool on;
    ProfScope(int k, double work, hipStream_t s) : kind(k), st(s), on(g_prof_on) { if (on) prof_begin(k, work, s); }
    ~ProfScope() { if (on) prof_end(kind, st); }
};

// ---------------------------------------------------------------------------------------------
// grouped "NT" GEMM:  out[r][c] = epi( sum_k A[r][k] * W[c][k] )   (nn.Linear without bias)
// A may be the concatenation [A | A2] along k (torch.cat([x, msg], 2) of transformero.py:91).
// ---------------------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_ROTARY = 2 };

struct GemmProblem {
    const float* A;     // [rows, K1] row-major, leading dimension lda
    const float* A2;    // [rows, K - K1] or nullptr
    const float* W;     // [ncols, K] row-major (the nn.Linear weight)
    float* out;         // [rows, ldo]
    const float* cosT;  // rotary tables [rows, C/2] (EPI_ROTARY)
    const float* sinT;
    const float* bias;  // [ncols] added before the activation (nn.Linear with bias) or nullptr
    const float* addend; // [rows, ldo] added last (e.g. tokens + embedding projection) or nullptr
    int rows, ncols, K, K1, lda, lda2, ldo;
    int epi;
    int rot_C;          // rotary: column c uses table index (c % rot_C) / 2
    float scale;        // applied last (1/sqrt(C) of matching.py:190)
    // strided batch (f32-MFMA kernels only): instance z = blockIdx.z uses A + z sA, W + z sW, out + z sO (floats)
    int nbatch;         // 0 or 1 = a single instance
    long long sA, sW, sO;
};

struct GemmBatch {
    GemmProblem p[4];
    int n;
};

int launch_gemm(const GemmBatch& g, hipStream_t st);
int gemm_configure();
void gemm_force_config(int c);

// ---------------------------------------------------------------------------------------------
// attention (transformero.py:79-85): segments of queries attending segments of keys
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q;       // [q_rows_total, ldq]   (rotary already applied)
    const float* k;       // [k_rows_total, ldk]
    const float* v;
    float* out;           // [q_rows_total, ldo]
    const uint8_t* qmask; // per q row (nullable = all valid)
    const uint8_t* kmask; // per k row
    int ldq, ldk, ldv, ldo;
    int H, d;             // heads, head dim (C = H*d)
    int nseg;             // number of segments
    // segment g: queries rows [q0 + g*qstride, +Lq), keys rows [k0 + g*kstride, +Lk)
    int q0, qstride, Lq, k0, kstride, Lk;
    // second family of segments (self attention over src AND tgt in one launch); nseg2 may be 0
    int nseg2, q0b, qstrideb, Lqb, k0b, kstrideb, Lkb;
    float scale;          // 1/sqrt(d)
    // plane-image output (pgemm.h) instead of `out`: rows < p_split go to pimg[0], the others (minus p_split) to pimg[1]; head h
    // occupies the k range [h p_dp, h p_dp + d) of the image (p_dp = d rounded up to 16, the pad is written as zeros);
    // bound of a query row = max over the segment's key rows of kbnd[row] * vnorm[0]  (|softmax V| <= max |V|), written to
    // pbnd[row] by the head-0 workgroups
    char* pimg[2]; int p_split, p_nct, p_dp; float* pbnd; const float* kbnd; const float* vnorm;
    // plane-image INPUT (attention_planes_kernel): q / k / v as images of p_nct chunks (head h at k = h p_dp), written by the
    // q|k|v GEMM's epilogue with the rotary embedding applied; same side split as pimg.  qbnd, kgb, vgb: bounds per token row;
    // all key rows of a segment carry ONE bound (the GEMM wrote the bound of the row's group: a pair's side), read at the
    // segment's first key row
    const char* qimg[2]; const char* kimg[2]; const char* vimg[2];
    const float* qbnd; const float* kgb; const float* vgb;
    int f16_single;       // plane-image form only: ONE fp16 product per contraction (hi planes only) -- the opt-in DR_LOOP_ATTN_F16 mode
    int xcd_groups;       // plane-image form only (set by the launcher): the query blocks of one (head, segment) run on ONE XCD -- they share its K / V
};
int launch_attention(const AttnArgs& a, hipStream_t st);
int attention_configure();
void attention_force_flash_min(int n);
void attention_force_split(int on);

// ---------------------------------------------------------------------------------------------
// row-wise ops
// ---------------------------------------------------------------------------------------------
// out[r] = (res ? res[r] : 0) + LayerNorm(x[r]) * g + b      (eps = 1e-5)
// rowmax (optional): max |out[r][:]| per row, for the consumer GEMM's operand scaling
int launch_layernorm(const float* x, int ldx, const float* g, const float* b, const float* res, int ldres, float* out,
                     int ldo, int rows, int C, hipStream_t st, float* rowmax = nullptr);
// post-LN form: out[r] = LayerNorm(x[r] + res[r]) * g + b   (vision3d AttentionLayer / AttentionOutput)
int launch_layernorm_postadd(const float* x, int ldx, const float* g, const float* b, const float* res, int ldres, float* out,
                             int ldo, int rows, int C, hipStream_t st);
// warped = R p + t per pair (R, t nullable), mean over the rows of each pair, Fourier embedding
// [p - mean | sin(2^l .), cos(2^l .)] of width (2L+1)*3 zero-padded to ldo (EXP/fusion_module.py:55-59)
int launch_fourier3d(const float* xyz, int P, int rows_per_pair, const float* R, const float* t, int L, float* emb, int ldo,
                     float* warped_ws, hipStream_t st);
// Fourier embedding of 2-D pixel coordinates, no centring (EXP/fusion_module.py:50-53)
int launch_fourier2d(const float* pix, int rows, int L, float* emb, int ldo, hipStream_t st);
// rotary tables of warped points: p' = R p + t (R,t per pair, nullable), cos/sin [rows, C/2]
int launch_vol_pe(const float* xyz, int rows, int rows_per_pair, const float* R, const float* t, int C, float ox,
                  float oy, float oz, float voxel, const float* freq, float* cosT, float* sinT, hipStream_t st, float* cs_pairs = nullptr);

// ---------------------------------------------------------------------------------------------
// top-K weighted Procrustes (procrustes.hip)
// ---------------------------------------------------------------------------------------------
// ws (optional, procrustes_workspace_bytes): lets tiles beyond 256 x 256 select their candidates with the whole chip
size_t procrustes_workspace_bytes(int P, int N, int M);
int launch_procrustes(const float* conf, const float* src_pcd, const float* tgt_pcd, const uint8_t* src_mask,
                      const uint8_t* tgt_mask, int P, int N, int M, int use_mask_len, float sample_rate, float max_cond,
                      float* R, float* t, float* Rf, float* tf, double* cond, int* ok, int* topk_idx, hipStream_t st,
                      void* ws = nullptr, size_t ws_bytes = 0);

// ---------------------------------------------------------------------------------------------
// diffusion-state kernels (stateops.hip)
// ---------------------------------------------------------------------------------------------
struct DdimArgs {
    double* x;               // [P, N*M] state, updated in place
    const float* x0;         // [P, N*M] x_start of this step
    const double* shift;     // [P] per-pair minimum or nullptr
    const float* noise;      // [P, N*M] xi of this step or nullptr
    const uint8_t* src_mask; // nullable
    const uint8_t* tgt_mask;
    int N, M, first_step;
    double sra, srm1, c, sigma;
    float sqrt_an;
};
// M, sm, tm (optional): minimum over the entries inside both masks only
// (scratch: pair_min_scratch_bytes(P) bytes whose last 64 P bytes -- the arrival counters -- are zero; enables the multi-workgroup form)
int launch_pair_min(const double* x, int P, int NM, double* out, hipStream_t st, int M = 0, const uint8_t* sm = nullptr,
                    const uint8_t* tm = nullptr, void* scratch = nullptr);
size_t pair_min_scratch_bytes(int P);
int launch_ddim(const DdimArgs& a, int P, hipStream_t st);
int launch_f32_to_f64(const float* in, double* out, size_t n, hipStream_t st);
int launch_f64_to_f32(const double* in, float* out, size_t n, hipStream_t st);
int launch_sigmoid(const double* in, double* out, size_t n, hipStream_t st);
// ws (optional, top1_union_workspace_bytes): arg-maxima with the whole chip (row blocks) instead of one workgroup per pair
size_t top1_union_workspace_bytes(int P, int N, int M, size_t elt);
template <typename T>
int launch_top1_union(const T* conf, int P, int N, int M, long long* out, int* count, hipStream_t st, const uint8_t* sm = nullptr,
                      const uint8_t* tm = nullptr, void* ws = nullptr, size_t ws_bytes = 0);

// collate.hip: in-place bitonic sort of (key, value) pairs ascending by (key, value); n_pad = a power of two (pad keys ~0ull sort last)
int launch_bitonic_sort(unsigned long long* keys, unsigned* vals, int n_pad, hipStream_t st);

// KPConv influence of a kernel point at squared distance d2 (blocks.py:304-321): mode & 3 = 0 'constant', 1 'linear', 2 'gaussian'
// (radius_gaussian with sigma = 0.3 extent, blocks.py:36-44); bit 2 of mode = aggregation 'closest' (applied by the caller: only the nearest
// kernel point of a neighbour keeps its influence, blocks.py:324-326)
enum { KP_CONSTANT = 0, KP_LINEAR = 1, KP_GAUSSIAN = 2, KP_CLOSEST = 4 };
__device__ __forceinline__ float kp_influence(int mode, float d2, float extent) {
    const int inf = mode & 3;
    if (inf == KP_LINEAR) return fmaxf(1.f - sqrtf(d2) / extent, 0.f);
    if (inf == KP_GAUSSIAN) { const float sig = extent * 0.3f; return expf(-d2 / (2.f * sig * sig + 1e-9f)); }
    return 1.f;
}

// sinkhorn.hip (internal form of dr_sinkhorn_*: `shift` = per-tile value subtracted first, nullable)
int sinkhorn_f32(int B, int N, int M, const float* scores, const uint8_t* sm, const uint8_t* tm, const float* bin_score,
                 int iters, int flags, float* out, void* ws, size_t ws_bytes, hipStream_t st, unsigned* call_status = nullptr);
int sinkhorn_f64(int B, int N, int M, const double* scores, const double* shift, const uint8_t* sm, const uint8_t* tm,
                 const float* bin_score, int iters, int flags, void* out, void* ws, size_t ws_bytes, hipStream_t st, unsigned* call_status = nullptr);
// call_status: the caller's own sticky status word (device, 4 bytes, zero-initialised by the caller): bit 0 = a co-resident Sinkhorn of THIS
// caller timed out (set beside the process-wide flag of dr_device_status, so that concurrent engines cannot swallow or misattribute a time-out)
int sinkhorn_call_status(unsigned* word, hipStream_t st, bool clear);

}  // namespace dr
