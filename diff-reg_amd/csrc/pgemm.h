// pgemm.h -- internal interface of the plane-image GEMM (pgemm.hip): the layer nn.Linears of the loop with BOTH operands
// pre-split into fp16 hi / lo planes in global memory, so that the kernel streams them by LDS-DMA only.
#pragma once
#include "kernels.h"

namespace dr {

// ---------------------------------------------------------------------------------------------------------------------
// Plane image of an activation matrix X [rows, K], K % 16 == 0:
//   [rb = row / 128][kc = k / 16][r = row % 128][64 bytes]   (a 128-row block of one 16-deep k-chunk is 8 KB contiguous: it
//   is copied to LDS by eight 1 KB DMA instructions and IS the LDS image the MFMA fragments are read from)
//   the 64 bytes of (row, kc) are four 16-byte units of 8 fp16: logical unit u = 0: hi k 0..7, 1: hi k 8..15, 2: lo k 0..7,
//   3: lo k 8..15, stored at position u ^ ((r >> 2) & 3) (conflict-free ds_read_b128 of a 32-row fragment)
//   values: hi = fp16(x 2^s), lo = fp16(x 2^s - hi) with s = 14 - floor(log2(bound[row])) (f16 range: |x| <= bound[row] is
//   the producer's promise; bound is an upper bound, not the maximum: fp16 keeps 22 significand bits for every element
//   within 2^-17 of it and 2^-40 bound absolutely below -- far inside the fp32 rounding of a dot product)
// Rows are padded to a multiple of 128 (the pad rows are never stored from).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PI_BM = 128;
inline size_t plane_image_bytes(size_t rows, int K) { return ((rows + PI_BM - 1) / PI_BM) * (size_t)(K / 16) * (PI_BM * 64); }

// view of `nblk` consecutive column blocks of a packed weight image (pgemm_pack_weights)
struct PgW {
    const char* img;     // [nblk][nct][BN rows][64 B]   (rows >= C of a block are zero)
    const float* cinv;   // [nblk][BN]  2^-s_c of every output column
    const float* wnorm;  // [nblk]      max_c sum_k |W[c][k]|  (output bound = input bound * wnorm)
    int nct;             // 16-deep k-chunks of the image (= those of the A operand)
    int sub;             // 0: one block = BN weight rows.  2: the WIDE-WAVE layout (pgemm16w_kernel): a logical block of up to 576 columns is two
                         //    sub-blocks of 288 weight rows, img [nblk][2][nct][288][64 B]; cinv [nblk][576] and wnorm [nblk] stay logical
};

enum { PG_F32 = 0, PG_PLANES = 1, PG_LN = 2 };

struct PgProblem {
    const char* A0; const char* A1;     // plane images of the A operand: [A0 | A1] along k (A1 may be null)
    const float* bnd0; const float* bnd1; // their per-row bounds
    int nc0, nc1;                       // 16-deep k-chunks of each
    PgW W;
    int nblk;                           // column blocks (of C columns each) this problem computes
    int rows, C;
    int k_alg;                          // algorithmic k (for the flop count of the profiler; 0 = 16 nct)
    int mode;
    // PG_F32: out[row][nb * blk_stride + col] = rot(acc) * scale     (PG_LN: optional fp32 copy of the result, nb = 0)
    float* out; int ldo; int blk_stride;
    const float* cosT; const float* sinT; int rot_mask; int rot_C; float scale;
    const float* csT;                   // optional: the same tables interleaved [rows][C/2][cos, sin] (one 16-byte load per rotated float4)
    int rot_piece_len, rot_piece_pad;   // head-padded output columns (pack out_len / out_pad): table index of column c' (0: identity)
    // PG_PLANES / PG_LN: plane image of the output, column block nb -> chunks p_kc0 + nb * (C / 16) ..
    char* pimg; int p_nct; int p_kc0; float* pbnd;
    long long pimg_blk_stride; long long pbnd_blk_stride;   // != 0: every column block writes its OWN image / bound array (q | k | v)
    // PG_PLANES bound of block nb: bit nb of grp_mask set -> grp_bnd[grp_first + row / grp_rows] instead of the row's own bound
    // (grp_bnd == nullptr with grp_mask != 0: the kernel takes the group maximum of bnd0 itself -- needs grp_rows % 128 == 0)
    const float* grp_bnd; int grp_mask, grp_first, grp_rows;
    int relu;
    // PG_LN: y = LayerNorm(acc) * gamma + beta (+ resid[row][col]);  bound = (bnd_res ? bnd_res[row] : 0) + *lnB
    //        ln_postadd (the vision3d layer, vision3d/layers/transformer.py:188-196, 262-271): y = LayerNorm(acc + resid) * gamma + beta;  bound = *lnB
    const float* gamma; const float* beta; const float* resid; int ldr; const float* bnd_res; const float* lnB;
    int ln_postadd;
    // nn.Linear bias (all modes): acc + bias[nb * C + col] before rotary / ReLU / LayerNorm; bias_max[nb] >= max |bias| of block nb
    // (added to the bound a PG_PLANES image is scaled by)
    const float* bias; const float* bias_max;
    // PG_LN launches of 64-row workgroups that would leave half the chip idle (launch_pgemm decides: `ksplit`): the k range of a row block is
    // SPLIT over two workgroups (ids 8 apart: the same XCD), each keeps 16 of a wave's 32 rows and hands the other 16 rows' partial sums to
    // its partner through xk_buf [tile][destination half][4 waves][9][2][64 lanes] float4 (sc1 accesses; the wide-wave kernel splits its 128 x 288 tiles the
    // same way: two of a wave's four rounds); xk_flags [tile][2] holds
    // the epoch of the last launch whose half has been written (zeroed by the owner of the workspace before the first launch; xk_epoch counts
    // the launches that use the buffer since then).  xk_status: the caller's sticky status word (bit 1 = a partner did not arrive; nullable).
    float* xk_buf; unsigned* xk_flags; unsigned xk_epoch; unsigned* xk_status; int ksplit;
    int xk_cap;                         // exchange units (one per split tile: 64-row block / 128 x 288 tile) this problem's xk_buf / xk_flags hold
};
constexpr int PG_XK_MAX_RB = 160;                                  // row blocks (of 64 rows) a split launch can have: xk_buf / xk_flags are sized for it
inline size_t pgemm_xk_buf_bytes(int bn) { return (size_t)PG_XK_MAX_RB * 2 * 4 * (bn / 64) * 2 * 64 * 16; }
inline size_t pgemm_xk_flag_bytes() { return (size_t)PG_XK_MAX_RB * 2 * sizeof(unsigned); }
struct PgBatch { PgProblem p[3]; int n; int dbg; };   // dbg (debug knob DR_PG_NOEPI): 1 = return behind the main loop (timing builds)

bool pgemm_shape_ok(int C);                      // column-block widths the kernel is built for
int pgemm_bn(int C);                             // rows of a packed weight block (256 for C <= 256, 448 for C <= 448, 576 above)
int launch_pgemm(const PgBatch& g, hipStream_t st);
int pgemm_configure();

// packed weights: nblk blocks of C rows of W [nblk * C, K] (row-major) -> image + cinv + wnorm.
// k order of the image: k' = (k / piece_len) * piece_pad + k % piece_len  (zero where k' % piece_pad >= piece_len);
// piece_len = K, piece_pad = K for the identity.  nct = ceil(K / piece_len) * piece_pad / 16.
size_t pgemm_weight_bytes(int C, int nblk, int nct);    // image + cinv + wnorm, 256-aligned
void pgemm_weight_view(void* buf, int C, int nblk, int nct, PgW* view);
int pgemm_pack_weights(const float* W, int nblk, int C, int K, int piece_len, int piece_pad, void* buf, hipStream_t st);
int pgemm_pack_weights_block(const float* W, int C, int K, int piece_len, int piece_pad, const PgW& view, int nb, hipStream_t st, int out_len = 0,
                             int out_pad = 0);
// the same for the wide-wave layout (PgW::sub = 2; C <= 576 columns per logical block; launches without LayerNorm only)
bool pgemm16w_shape_ok(int C);
size_t pgemm16w_weight_bytes(int nblk, int nct);
void pgemm16w_weight_view(void* buf, int nblk, int nct, PgW* view);
int pgemm16w_pack_weights_block(const float* W, int C, int K, int piece_len, int piece_pad, const PgW& view, int nb, hipStream_t st, int out_len = 0,
                                int out_pad = 0);
int launch_group_max(const float* bnd, int ngroups, int grp_rows, float* out, hipStream_t st);

// fp32 rows -> plane image with bound[row] = max |x[row][:]| (the external features entering the first layer)
int launch_planes_from_f32(const float* x, int ldx, int rows, int K, char* img, float* bnd, hipStream_t st, const float* bnd_in = nullptr);
int launch_planes_to_f32(const char* img, const float* bnd, int rows, int K, float* out, int ldo, hipStream_t st);
// *out = sqrt(C) max|gamma| + max|beta|: an upper bound of |LayerNorm(.) gamma + beta|
int launch_ln_bound(const float* gamma, const float* beta, int C, float* out, hipStream_t st);
// out[b] = max |x[b * n .. b * n + n - 1]| * (1 + 1e-4), b < nblk  (bias_max of a stacked bias vector)
int launch_absmax_blocks(const float* x, int nblk, int n, float* out, hipStream_t st);

}  // namespace dr
