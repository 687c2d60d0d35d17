/* diffreg_hip.h -- C ABI of libdiffreg_hip.so (MI355X / gfx950).
 *
 * The reference (wuqianliang/Diff-Reg) has no FFI on this path: the boundary is a set of Python
 * functions/classes.  Each entry point below replaces the body of one of
 * them; the Python side (the modules under diff-reg_amd/models) keeps the reference signatures and binds these
 * with ctypes.  Citations: 3D/ = Diff-Reg-3dmatch/, 4D/ = Diff-Reg-4dmatch/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with `h_`; tensors are contiguous
 *     row-major; the caller owns every buffer, workspaces are passed in explicitly;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on it and never
 *     synchronise the host;
 *   - return 0 on success, a negative DR_E* code otherwise (dr_strerror gives the text);
 *   - "pairs": P independent scene pairs (reference inference is B = 1 per pair); per-pair
 *     quantities such as x.min() are per pair.
 */
#ifndef DIFFREG_HIP_H
#define DIFFREG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_OK 0
#define DR_EINVAL (-1)   /* bad argument (size, NULL pointer, unsupported flag)            */
#define DR_ELAUNCH (-2)  /* a HIP launch or API call failed (dr_last_hip_error has detail) */
#define DR_ENOSUP (-3)   /* shape not supported by this build                               */
#define DR_EWORKSPACE (-4) /* workspace too small                                           */
#define DR_ETIMEOUT (-5) /* a kernel gave up waiting for another workgroup (dr_device_status) */

/* ABI version of THIS header.  dr_version() returns the library's; a binding must refuse to run when they differ in major or
 * minor (the Python mirror does, diffreg_hip/lib.py).  History of breaks:
 *   0.2.0  dr_loop_trace grew the teacher-forcing fields (a 0.1.0 caller's struct is too short); dr_procrustes_f32 and
 *          dr_top1_union_f32 / _f64 take (workspace, workspace_bytes) in front of `stream` since the last 0.1.0 builds -- a caller
 *          compiled against the header without them passes its stream in the workspace slot. */
#define DR_ABI_VERSION 202
int dr_version(void);                 /* major*10000 + minor*100 + patch */
const char* dr_strerror(int code);
const char* dr_last_hip_error(void);  /* text of the last failing HIP call on this thread */

/* Device-side failures.  Every call above returns when its kernels are ENQUEUED, so a failure that only a running kernel can
 * detect cannot come back as a return value.  There is one such failure: the single-launch Sinkhorn of tiles beyond 256 x 256
 * exchanges column sums between workgroups that must all be resident; its spins are bounded, and a workgroup that gives up
 * writes NaN to every output entry it owns and sets a sticky, process-wide flag on the device.  The outputs of such a call are
 * UNSPECIFIED as a whole: a sibling workgroup that was scheduled late may have summed partials of peers that had moved on and
 * written finite but wrong values -- the flag, not the data, is the signal; whoever reads it first (with `clear`) consumes it.
 * dr_device_status is the ONE entry point that synchronises: it waits for `stream`, reads the flag (clearing it when `clear`
 * is non-zero) and returns DR_OK or DR_ETIMEOUT.  The host mirrors call it wherever they already synchronise to read a match
 * count.  (The launcher checks residency with the occupancy API and takes the multi-launch form when the launch would not
 * fit beside a second one, so the flag means something else holds the CUs: a foreign kernel, a CU mask, a partitioned device.) */
int dr_device_status(void* stream, int clear);

/* ---------------------------------------------------------------------------------------------
 * Sinkhorn with dustbins.  Replaces log_optimal_transport + exp + [:-1,:-1] slice
 * (3D/models/matching.py:61-93 and its call sites matching.py:207-216, pipeline.py:264-277,
 * pipeline.py:293-302) for B independent N x M tiles.
 *   scores      [B,N,M]   (f32 or f64 entry point)
 *   src_mask    [B,N] uint8 (torch.bool) or NULL = all valid;  tgt_mask [B,M] likewise
 *   bin_score   device pointer to ONE float (the nn.Parameter `bin_score`)
 *   iters       Sinkhorn iterations (3 everywhere in the reference)
 *   out         DR_SK_OUT_CONF: [B,N,M] = exp(logZ)[:, :-1, :-1];  DR_SK_OUT_LOG: [B,N+1,M+1] logZ
 *   workspace   dr_sinkhorn_workspace_bytes(...) bytes (0 for the register-resident f32 path)
 * flags:
 *   DR_SK_MINSHIFT    subtract the per-tile minimum first (pipeline.py:239,264 `x - x.min()`)
 *   DR_SK_APPLY_MASK  treat entries outside src_mask x tgt_mask as -inf (the masked_fill_ of
 *                     pipeline.py:296 / matching.py:209-211), whatever the buffer holds
 *   DR_SK_OUT_F32     (f64 entry point) write `out` as float -- the `.type(torch.float32)` of
 *                     pipeline.py:302
 *   DR_SK_STRICT      compute in the input dtype with the streaming kernel (fp64 state stays
 *                     fp64 end to end); default: scaling-form fp32 arithmetic on row-max-shifted
 *                     exponentials held in registers
 * Marginals follow the reference with masks: every padded row/column keeps mass (quirk Q19).
 */
#define DR_SK_OUT_CONF 0x0
#define DR_SK_OUT_LOG 0x1
#define DR_SK_MINSHIFT 0x2
#define DR_SK_APPLY_MASK 0x4
#define DR_SK_OUT_F32 0x8
#define DR_SK_STRICT 0x10
/* DR_SK_RAGGED (with masks): rows / columns outside the masks do not exist -- no marginal mass, no dustbin share --
 * so a padded tile gives exactly the result of its unpadded (ms x ns) problem.  Without it padded rows / columns
 * keep their marginal mass like the reference's batched call does (quirk Q19), which is NOT the B = 1 result. */
#define DR_SK_RAGGED 0x20

size_t dr_sinkhorn_workspace_bytes(int B, int N, int M, int elem_bytes, int flags);

int dr_sinkhorn_f32(int B, int N, int M, const float* scores, const uint8_t* src_mask,
                    const uint8_t* tgt_mask, const float* bin_score, int iters, int flags,
                    float* out, void* workspace, size_t workspace_bytes, void* stream);

/* OPT-IN reduced precision (SURVEY 8b lists dr_sinkhorn_f16; never used by the loop): the score tiles and the confidences are stored as
 * IEEE fp16 -- half the bytes of an HBM-bound op -- the iteration itself runs in fp32 registers exactly like dr_sinkhorn_f32.  Tiles of up to
 * 256 x 256 (the register-resident kernel: BASELINE cfg1 / cfg2 sizes; DR_ENOSUP beyond); flags as dr_sinkhorn_f32 except DR_SK_STRICT /
 * DR_SK_OUT_LOG.  Outside the 1e-4 contract by construction: an fp16 confidence carries 11 bits. */
int dr_sinkhorn_f16(int B, int N, int M, const void* scores_f16, const uint8_t* src_mask, const uint8_t* tgt_mask, const float* bin_score,
                    int iters, int flags, void* out_f16, void* stream);
int dr_sinkhorn_f64(int B, int N, int M, const double* scores, const uint8_t* src_mask,
                    const uint8_t* tgt_mask, const float* bin_score, int iters, int flags,
                    void* out /* double*, or float* with DR_SK_OUT_F32 */, void* workspace,
                    size_t workspace_bytes, void* stream);


/* ---------------------------------------------------------------------------------------------
 * dr_init: sets kernel attributes (dynamic LDS sizes).  Call once per process before the first
 * launch and before any stream capture (the Python loader does).
 */
int dr_init(void);

/* Per-kernel-family timing with HIP events recorded on the launch stream, for bench.py's roofline
 * line.  Families: 0 gemm (f32-input MFMA kernels), 1 attention, 2 layernorm, 3 position code, 4 sinkhorn,
 * 5 procrustes, 6 state (min / ddim / read-out), 7 gemm_split (packed-weight split-operand kernels).  work[k] = algorithmic FLOPs (gemm, attention) or bytes (others)
 * summed over the recorded launches.  Do not enable during a stream capture. */
#define DR_PROF_KINDS 8
void dr_prof_enable(int on);
int dr_prof_collect(int* calls, double* ms, double* work);

/* ---------------------------------------------------------------------------------------------
 * VolumetricPositionEncoding.forward (3D/models/position_encoding.py:49-87) of `rows` points,
 * optionally warped first by a per-pair rigid motion  p' = R p + t  (3D/models/pipeline.py:306).
 *   xyz [rows,3]; R [rows/rows_per_pair, 9] row-major and t [.., 3], both NULL for no warp
 *   freq [C/6] = exp(arange(0, C/3, 2) * (-ln 1e4 / (C/3)))  (device; the caller computes it so the
 *               table is bit-identical to torch's)
 *   cos_out, sin_out [rows, C/2]: entry a*(C/6)+k is the angle of channels 2(a*C/6+k) and +1
 *               (the reference stores each angle twice: position_encoding.py:74-79)
 */
int dr_vol_pe_f32(int rows, int rows_per_pair, int C, const float* xyz, const float* R, const float* t,
                  float origin_x, float origin_y, float origin_z, float voxel, const float* freq,
                  float* cos_out, float* sin_out, void* stream);

/* nn.Linear without bias (+ optional rotary / ReLU / scale epilogue): out[rows,ncols] = x W^T.
 * epilogue bits: 1 = ReLU, 2 = rotary with cos/sin [rows, rot_C/2] (embed_rotary,
 * position_encoding.py:25-35).  Used by the parity tests of the GEMM kernel. */
int dr_linear_f32(int rows, int ncols, int K, const float* x, const float* W, float* out, int epilogue,
                  const float* cos_t, const float* sin_t, int rot_C, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Plane images: the layer nn.Linears of the loop with BOTH operands pre-split into fp16 hi / lo planes (the
 * default path of dr_denoise_loop for C <= 576, C % 16 == 0; 3D/models/transformero.py:26-96; column-block geometries of
 * 256 / 448 / 576 columns: C <= 256 -- the 2D-3D layer -- packs 256-row weight blocks).
 * Image of X [rows, K] (K % 16 == 0, rows padded to a multiple of 128): [row / 128][k / 16][row % 128][64 bytes], the
 * 64 bytes = 16-byte units (hi k 0..7 | hi k 8..15 | lo k 0..7 | lo k 8..15) stored at unit ^ ((row >> 2) & 3);
 * hi = fp16(x 2^s), lo = fp16(x 2^s - hi), s = 14 - floor(log2(bound[row])) with bound[row] >= max |x[row][:]| (an
 * upper bound that producers propagate analytically, so nobody sweeps a row for its maximum).  x = (hi + lo) 2^-s
 * keeps 22 significand bits for |x| >= 2^-17 bound and is rounded to 2^-40 bound absolutely below.
 * A 128-row block of one k-chunk is 8 KB contiguous and is the LDS image of the GEMM: operands stream by LDS-DMA.
 */
size_t dr_plane_image_bytes(int rows, int K);
/* fp32 rows -> image, bound[row] = max |x[row][:]| (the external features entering the first layer) */
int dr_planes_from_f32(int rows, int K, const float* x, int ldx, void* image, float* bound, void* stream);
/* the same with the caller's bounds (bound_in[row] >= max |x[row][:]|; e.g. one bound for all the key rows of a pair) */
int dr_planes_from_f32_bounded(int rows, int K, const float* x, int ldx, const float* bound_in, void* image, float* bound, void* stream);
/* einsum / mask / softmax / einsum of GeometryAttentionLayer.forward (3D/models/transformero.py:79-85) on plane images: P segments of
 * Lq queries attending Lk keys (rows p Lq .. / p Lk .. of the images), H heads of d features laid out head-padded (head h at
 * k = h dp, dp = d rounded up to 16, zeros in the pad; images of H dp columns).  All key rows of a segment must carry ONE k bound
 * and ONE v bound (read at the segment's first key row).  Three fp16 MFMA products per fp32 product, K / V tiles by LDS-DMA.
 * out_image: the merge projection's operand (same layout), out_bound [P Lq].  d in {64, 108, 132} (dp 64, 112, 144). */
int dr_attention_planes(int P, int Lq, int Lk, int H, int d, const void* q_image, const float* q_bound, const void* k_image,
                        const float* k_bound, const void* v_image, const float* v_bound, const uint8_t* q_mask, const uint8_t* k_mask,
                        void* out_image, float* out_bound, void* stream);
/* OPT-IN reduced precision: the same op with ONE fp16 product per contraction (the hi planes of q, k, v and of P only) instead of three --
 * what dr_loop_config.flags & DR_LOOP_ATTN_F16 selects inside the loops.  11-bit operands: outside the 1e-4 contract, never a default. */
int dr_attention_planes_f16(int P, int Lq, int Lk, int H, int d, const void* q_image, const float* q_bound, const void* k_image,
                            const float* k_bound, const void* v_image, const float* v_bound, const uint8_t* q_mask, const uint8_t* k_mask,
                            void* out_image, float* out_bound, void* stream);
/* image -> fp32 rows (tests) */
int dr_planes_to_f32(int rows, int K, const void* image, const float* bound, float* out, int ldo, void* stream);
/* weights W [nblk * C, K] (nn.Linear layout; nblk stacked layers of C output columns each, C <= 448, C % 16 == 0) ->
 * packed image + per-column scales + per-block L1 norms.  The k order of the image is
 * k' = (k / piece_len) * piece_pad + k % piece_len (zeros where k' % piece_pad >= piece_len): piece_len = piece_pad = K for
 * the identity; (d, round_up(d, 16)) for the merge projection behind the attention kernel's head-padded image. */
size_t dr_plane_weight_bytes(int nblk, int C, int K, int piece_len, int piece_pad);
int dr_pack_weight_planes_f32(int nblk, int C, int K, int piece_len, int piece_pad, const float* W, void* packed, void* stream);
/* The same weights in the WIDE-WAVE layout (round 6; C <= 576, C % 16 == 0): every block of C output columns as two sub-blocks of 288 weight
 * rows -- what the 128 x 288 workgroups of the DR_PL_F32 / DR_PL_PLANES launches of the 576-column geometry stream (4DMatch: C = 528, heads padded
 * to 144).  Pass the buffer with dr_planes_linear.weight_layout = DR_PL_LAYOUT_WIDE; DR_PL_LN launches take the block layout only. */
size_t dr_plane_weight_bytes_wide(int nblk, int C, int K, int piece_len, int piece_pad);
int dr_pack_weight_planes_wide_f32(int nblk, int C, int K, int piece_len, int piece_pad, const float* W, void* packed, void* stream);
/* *out = (sqrt(C) max|gamma| + max|beta|): upper bound of |LayerNorm(.) gamma + beta| (device scalar) */
int dr_ln_bound_f32(int C, const float* gamma, const float* beta, float* out, void* stream);

#define DR_PL_F32 0     /* out[row][nb * blk_stride + c] = rotary(acc) * scale                                      */
#define DR_PL_PLANES 1  /* out_image (+ out_bound) = [relu](acc); bound = max(bound0, bound1) * ||W_nb||            */
#define DR_PL_LN 2      /* y = LayerNorm(acc) gamma + beta [+ resid]; -> out (optional) and out_image (optional)    */
typedef struct {
    int rows, C, nblk;             /* out columns = nblk blocks of C                                                  */
    const void* a0; const float* bound0; int k0;   /* A operand = [a0 | a1] along k (images + their bounds)           */
    const void* a1; const float* bound1; int k1;   /* a1 may be NULL                                                  */
    const void* packed;            /* dr_pack_weight_planes_f32 of W [nblk * C, k0 + k1] (same k order as the images)  */
    int mode;                      /* DR_PL_*                                                                         */
    float* out; int ldo; int blk_stride;
    const float* cos_t; const float* sin_t; int rot_mask; int rot_C; float scale;   /* bit nb of rot_mask: rotary      */
    void* out_image; int out_image_k; int out_k0; float* out_bound;   /* block nb -> columns out_k0 + nb * C ..        */
    int relu;
    const float* gamma; const float* beta; const float* resid; int ldr; const float* bound_resid; const float* ln_bound;
    /* nn.Linear bias [nblk * C] (NULL = none), added before rotary / ReLU / LayerNorm; bias_max [nblk] (device, dr_bias_max_f32):
     * an upper bound of |bias| per block, needed with DR_PL_PLANES (it enters the bound the image is scaled by) */
    const float* bias; const float* bias_max;
    /* DR_PL_LN with resid: 0 = y = LayerNorm(acc) gamma + beta + resid (3D/models/transformero.py:94-96);
     *                      1 = y = LayerNorm(acc + resid) gamma + beta (the vision3d layer, vision3d/layers/transformer.py:188-196, 262-271) */
    int ln_postadd;
    /* DR_PL_PLANES: `out` (optional) also receives the block as fp32 rows (ldo, blk_stride as in DR_PL_F32) */
    int weight_layout;             /* DR_PL_LAYOUT_*: how `packed` was packed (ABI 0.2.1)                                */
    /* ABI 0.2.2, optional (NULL = never): dr_plane_split_workspace_bytes(C) bytes of device memory.  With it, a launch that would fill at most
     * half the chip (a DR_PL_LN launch of 64-row workgroups; a DR_PL_LAYOUT_WIDE launch of 128 x 288 tiles) splits every tile's k range over
     * TWO workgroups that swap half of their partial sums (the loops do this with their own workspace).  The first word of the workspace is a
     * status word: dr_plane_split_status() -> DR_ETIMEOUT if a workgroup's partner never arrived (its rows are NaN then). */
    void* split_workspace; size_t split_workspace_bytes;
} dr_planes_linear;
size_t dr_plane_split_workspace_bytes(int C);
int dr_plane_split_status(void* split_workspace, void* stream, int clear);
#define DR_PL_LAYOUT_BLOCK 0   /* dr_pack_weight_planes_f32                                                           */
#define DR_PL_LAYOUT_WIDE 1    /* dr_pack_weight_planes_wide_f32 (DR_PL_F32 / DR_PL_PLANES only)                      */
int dr_linear_planes_f32(const dr_planes_linear* args, void* stream);
/* out[b] = max |bias[b n .. b n + n - 1]| (1 + 1e-4), b < nblk */
int dr_bias_max_f32(int nblk, int n, const float* bias, float* out, void* stream);

/* strided batch of the same product: out[z] [rows, ncols] = A[z] [rows, K] W[z]^T (W[z] [ncols, K]) * scale, z < nbatch, operands contiguous with the
 * given strides (floats); K % 4 == 0.  The N x M similarity of a batch of pairs (matching.py:190-196); the per-head products of the training backward. */
int dr_gemm_nt_batched_f32(int nbatch, int rows, int ncols, int K, const float* A, long long stride_a, const float* W, long long stride_w, float* out,
                           long long stride_o, float scale, void* stream);

/* nn.Linear with a bias and explicit leading dimensions (x [rows, lda], out [rows, ldo]): the 1x1 Conv1d `coarse_out`
 * of the backbone (3D/models/backbone.py:66, 155-156) and its UnaryBlocks (bias = NULL). */
int dr_linear_ex_f32(int rows, int ncols, int K, const float* x, int lda, const float* W, const float* bias, float* out,
                     int ldo, int epilogue, float scale, void* stream);

/* ---- KPFCN backbone ops (3D/models/blocks.py) -------------------------------------------------------
 * dr_kpconv_gather_f32: the gather / influence / neighbour-reduction half of KPConv.forward (blocks.py:288-375) with
 *   the neighbour-count normalisation (blocks.py:390-393) folded in:
 *     weighted[q][k*Cin + c] = (sum_h max(0, 1 - |s[nb[q][h]] - q_q - kp_k| / extent) * x[nb[q][h]][c]) / num_q
 *   neighb_inds [Nq,H] int64 with the shadow index Ns (zero features, point at +1e6); num_q = max(1, #neighbours whose
 *   feature sum is > 0).  The other half is ONE dr_linear_*: out = weighted @ W2^T with W2[co][k*Cin + c] =
 *   weights[k][c][co] (the reference multiplies per kernel point and sums over K, blocks.py:382-387).
 *   ld_weighted >= K*Cin (padding columns are zeroed).  K <= 16, H <= 64.
 * dr_col_stats_f32 / dr_norm_apply_f32: BatchNormBlock with use_bn = InstanceNorm1d over the points of the stacked cloud
 *   per channel, no affine, biased variance, eps 1e-5 (blocks.py:430-446): mean / rstd per column, then
 *     out = act( (a - mean_a) rstd_a + [ (b - mean_b) rstd_b  |  b  |  0 ] ),  act = LeakyReLU(leaky_slope) or none
 *   (UnaryBlock blocks.py:479-484; the residual sum of ResnetBottleneckBlock blocks.py:650-660).
 * dr_gather_pool_f32: max_pool (first_only = 0) / closest_pool (first_only = 1) of blocks.py:56-87; indices >= n1 read
 *   a zero row. */
int dr_kpconv_gather_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts,
                         const int64_t* neighb_inds, const float* x, const float* kernel_points, float extent,
                         float* weighted, int ld_weighted, void* stream);
/* the same with the options of KPConv no shipped yaml selects (blocks.py:304-326; ABI 0.2.1): influence DR_KP_CONSTANT (every neighbour weighs 1),
 * DR_KP_LINEAR (dr_kpconv_gather_f32) or DR_KP_GAUSSIAN (exp(-d^2 / (2 (0.3 extent)^2 + 1e-9))); closest != 0 = aggregation_mode 'closest': a
 * neighbour contributes through its nearest kernel point only (first minimum, as torch.argmin). */
#define DR_KP_CONSTANT 0
#define DR_KP_LINEAR 1
#define DR_KP_GAUSSIAN 2
int dr_kpconv_gather_mode_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                              const float* x, const float* kernel_points, float extent, int influence, int closest, float* weighted,
                              int ld_weighted, void* stream);
size_t dr_col_stats_workspace_bytes(int N, int C);
int dr_col_stats_f32(int N, int C, const float* x, int ldx, float* mean, float* rstd, void* workspace,
                     size_t workspace_bytes, void* stream);
int dr_norm_apply_f32(int N, int C, const float* a, int lda, const float* mean_a, const float* rstd_a, const float* b,
                      int ldb, const float* mean_b, const float* rstd_b, float leaky_slope, int activate, float* out,
                      int ldo, void* stream);
int dr_gather_pool_f32(int n2, int H, int ld_inds, int d, const float* x, int n1, const int64_t* inds, int first_only,
                       float* out, void* stream);

/* Backward of the three ops above (the backbone's training path; the nn.Linear halves are products on dr_linear_*):
 * dr_kpconv_gather_backward_f32: grad_x [Ns, Cin] (zeroed here, then fp32 atomics) from grad_weighted [Nq, ld_weighted]; kernel points and
 *   point positions carry no gradient (the shipped configuration: rigid kernels), the neighbour count is piecewise constant.
 * dr_norm_backward_f32: out = act(xhat_a + [xhat_b | b | 0]) as dr_norm_apply_f32 computed it -> grad_a (and grad_b when b took part);
 *   `out` is the forward result (the LeakyReLU's derivative is read off its sign).  Two float64 column reductions over a fixed grid.
 * dr_gather_pool_backward_f32: the gradient of a pooled row goes to its first maximal neighbour (torch.max) / its first neighbour. */
int dr_kpconv_gather_backward_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                                  const float* x, const float* kernel_points, float extent, const float* grad_weighted, int ld_weighted,
                                  float* grad_x, void* stream);
int dr_kpconv_gather_backward_mode_f32(int Nq, int Ns, int H, int Cin, int K, const float* q_pts, const float* s_pts, const int64_t* neighb_inds,
                                       const float* x, const float* kernel_points, float extent, int influence, int closest,
                                       const float* grad_weighted, int ld_weighted, float* grad_x, void* stream);
size_t dr_norm_backward_workspace_bytes(int N, int C);
int dr_norm_backward_f32(int N, int C, const float* grad_out, int ldg, const float* out, int ldo, const float* a, int lda, const float* mean_a,
                         const float* rstd_a, const float* b, int ldb, const float* mean_b, const float* rstd_b, float leaky_slope, int activate,
                         float* grad_a, int ldga, float* grad_b, int ldgb, void* workspace, size_t workspace_bytes, void* stream);
int dr_gather_pool_backward_f32(int n2, int H, int ld_inds, int d, const float* x, int n1, const int64_t* inds, int first_only, const float* grad_out,
                                float* grad_x, void* stream);

/* weights of one GeometryAttentionLayer in the reference state-dict layout ([out,in] row-major;
 * 3D/models/transformero.py:26-41): host struct of device pointers */
typedef struct {
    const float *q_proj, *k_proj, *v_proj, *merge; /* [C,C]          */
    const float *mlp0;                             /* [2C,2C]        */
    const float *mlp2;                             /* [C,2C]         */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [C]       */
} dr_layer_weights;

/* GeometryAttentionLayer.forward(x, source, x_pe, source_pe, x_mask, source_mask)
 * (3D/models/transformero.py:43-96) for P independent pairs:
 *   x [P*Lx, C], y [P*Ly, C]; cos/sin tables [P*Lx, C/2] and [P*Ly, C/2]; masks uint8 or NULL
 *   out [P*Lx, C] = x + LN2(mlp([x, LN1(merge(attn))]))
 *   workspace: dr_attention_layer_workspace_bytes(P, Lx, Ly, C)
 */
size_t dr_attention_layer_workspace_bytes(int P, int Lx, int Ly, int C);
int dr_attention_layer_f32(const dr_layer_weights* w, int C, int H, int P, int Lx, int Ly, const float* x,
                           const float* y, const float* cos_x, const float* sin_x, const float* cos_y,
                           const float* sin_y, const uint8_t* x_mask, const uint8_t* y_mask, float* out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same layer for the configuration branches no shipped yaml selects (3D/models/transformero.py:50-57, 246-252;
 * configs/test/3dmatch.yaml:45 "options: ['rotary', 'sinusoidal']", :1 "entangled"):
 *   cos_x .. sin_y all NULL : no rotary code on q and k -- pe_type 'sinusoidal', or a layer called without position codes (entangled = True);
 *   xq [P*Lx, C], yk [P*Ly, C] (both or neither; only without rotary tables): the inputs of the q and k projections where they are not x and y --
 *     the sinusoidal form q = W_q (x + pe_x), k = W_k (y + pe_y), v = W_v y; the residual and the mlp's cat[x, message] keep x.
 * Same workspace.  dr_attention_layer_f32 is this entry with xq = yk = NULL and all four tables given. */
int dr_attention_layer_pe_f32(const dr_layer_weights* w, int C, int H, int P, int Lx, int Ly, const float* x, const float* y,
                              const float* xq, const float* yk, const float* cos_x, const float* sin_x, const float* cos_y,
                              const float* sin_y, const uint8_t* x_mask, const uint8_t* y_mask, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* The same layer for TRAINING (what torch autograd records through transformero.py:43-96): the forward that keeps what its
 * backward needs, and the whole backward, one call each -- every kernel launched inside the library (the per-op entries further down drive the same
 * kernels one by one).  x [B*L, C], y [B*S, C] contiguous; rotary tables [B*L, C/2] / [B*S, C/2]; masks uint8 [B*L] / [B*S] or both NULL.
 *   forward : out [B*L, C]; `saved` = caller memory of dr_attention_layer_train_saved_bytes(B, L, S, C): q | k (rotary applied) | v | heads' output |
 *             merge output | norm1 output | hidden activation | mlp output | (mean, rstd) of both LayerNorms -- opaque to the caller, handed to the backward
 *   backward: grad_out [B*L, C] -> grad_x [B*L, C], grad_y [B*S, C] (separate buffers also when x == y: the caller adds them) and the ten parameter
 *             gradients (dr_layer_grads: the layout of dr_layer_weights; overwritten, not accumulated).  Position codes are constants of the graph
 *             (the reference detaches them, position_encoding.py:83-84).  workspace: dr_attention_layer_backward_workspace_bytes(B, H, L, S, C). */
typedef struct {
    float *q_proj, *k_proj, *v_proj, *merge; /* [C,C]          */
    float *mlp0;                             /* [2C,2C]        */
    float *mlp2;                             /* [C,2C]         */
    float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [C]       */
} dr_layer_grads;
size_t dr_attention_layer_train_saved_bytes(int B, int L, int S, int C);
int dr_attention_layer_train_forward_f32(const dr_layer_weights* w, int C, int H, int B, int L, int S, const float* x, const float* y,
                                         const float* cos_x, const float* sin_x, const float* cos_y, const float* sin_y,
                                         const uint8_t* x_mask, const uint8_t* y_mask, float* out, void* saved, size_t saved_bytes, void* stream);
size_t dr_attention_layer_backward_workspace_bytes(int B, int H, int L, int S, int C);
int dr_attention_layer_backward_f32(const dr_layer_weights* w, int C, int H, int B, int L, int S, const float* x, const float* y,
                                    const float* cos_x, const float* sin_x, const float* cos_y, const float* sin_y, const uint8_t* x_mask,
                                    const uint8_t* y_mask, const void* saved, const float* grad_out, float* grad_x, float* grad_y,
                                    const dr_layer_grads* grads, void* workspace, size_t workspace_bytes, void* stream);

/* SoftProcrustesLayer.forward (3D/models/procrustes.py:48-93): top-K of conf, weighted Kabsch,
 * fp64 3x3 SVD ON DEVICE (replaces the .cpu().double().svd() of procrustes.py:35-36), gate.
 *   conf [P,N,M] float32; src_pcd [P,N,3]; tgt_pcd [P,M,3]
 *   use_mask_len: 0 = K from the padded sizes (3D, 2D3D), 1 = K from the mask sums (4D variant,
 *                 4D/models/procrustes.py:61-62)
 *   R,t = solution; R_forwd,t_forwd = solution or identity when cond >= max_condition_num;
 *   topk_idx (optional, [P,K]) flat indices i*M+j of the selected entries (a set: unordered)
 *   Ties at the K-th value: the lowest flat indices are taken (torch.topk leaves that choice implementation-defined).
 *   workspace: dr_procrustes_workspace_bytes(P, N, M) bytes (0 for tiles up to 256 x 256: pass NULL); tiles beyond that select
 *   with the whole chip and need it -- DR_EWORKSPACE when it is missing or too small.
 */
size_t dr_procrustes_workspace_bytes(int P, int N, int M);
int dr_procrustes_f32(int P, int N, int M, const float* conf, const float* src_pcd, const float* tgt_pcd,
                      const uint8_t* src_mask, const uint8_t* tgt_mask, int use_mask_len, float sample_rate,
                      float max_condition_num, float* R, float* t, float* R_forwd, float* t_forwd,
                      double* condition, int32_t* solution_mask, int32_t* topk_idx, void* workspace, size_t workspace_bytes,
                      void* stream);

/* d loss / d conf of the fit above (4DMatch trains its L1 motion term through (R, t), 3D/models/loss.py:108-128): the K
 * entries dr_procrustes_f32 selected (topk_idx [P,K]) are the fit's weights, everything else gets 0.  grad_R [P,9], grad_t [P,3] are the
 * gradients of R and t (for R_forwd / t_forwd add them where solution_mask is set); k_count [P] (optional): entries of a pair that carry
 * weight (the 4D variant's K from the mask sums).  The adjoint of the 3 x 3 SVD in float64 on the device, replacing the reference's
 * autograd path through `Sxy.cpu().double().svd()` (procrustes.py:35-36).  grad_conf [P,N,M] is zeroed here. */
int dr_procrustes_backward_f32(int P, int N, int M, int K, const float* conf, const float* src_pcd, const float* tgt_pcd, const int32_t* topk_idx,
                               const int32_t* k_count, const float* grad_R, const float* grad_t, float* grad_conf, void* stream);

/* Diagnostics (kernel-forcing setters, phase stamps, the environment switch) are NOT part of the drop-in boundary: they are
 * declared in include/diffreg_hip_debug.h.  The library reads no environment variable unless dr_debug_enable_env(1) was called. */

/* mutual_topk_select(conf, k=1, largest=True, threshold=None, mutual=False) + the [0,i,j] rows of
 * 3D/models/pipeline.py:275-278.  matches [P, N+M, 3] int64 (first count[p] rows valid), ascending (i, j); the first
 * occurrence wins an arg-max tie.  N + M <= 4096.  From 32 rows on the arg-maxima come from row-block workgroups over the
 * whole chip, which need dr_top1_union_workspace_bytes(P, N, M, elem_bytes) bytes of workspace (elem_bytes 4 / 8; 0 = none
 * needed, pass NULL); DR_EWORKSPACE when it is missing or too small. */
size_t dr_top1_union_workspace_bytes(int P, int N, int M, int elem_bytes);
int dr_top1_union_f64(int P, int N, int M, const double* conf, int64_t* matches, int32_t* count, void* workspace,
                      size_t workspace_bytes, void* stream);
int dr_top1_union_f32(int P, int N, int M, const float* conf, int64_t* matches, int32_t* count, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Pipeline.split_feats (3D/models/pipeline.py:350-379): dst[dst_index[i]][:] = src[src_index[i]][:], i < n, rows of C floats
 * (the stacked coarse features / points of the backbone scattered into the zero-padded [B * N_max, C] tensors).
 * src has n_src_rows rows, dst n_dst_rows.  Indices behave like torch's indexed assignment: values in [-rows, -1] wrap; any other
 * out-of-range index is skipped (no access) and *status (device int32, optional, zeroed by the caller) is set to 1 -- where
 * PyTorch raises IndexError; the Python mirror reads the flag and raises. */
int dr_scatter_rows_f32(int n, int C, const float* src, int64_t n_src_rows, const int64_t* src_index, const int64_t* dst_index, float* dst,
                        int64_t n_dst_rows, int32_t* status, void* stream);

/* Matching.get_match(conf, thr, mutual) (3D/models/matching.py:126-143; what 4D/lib/tester.py:266 applies to conf_matrix_pred
 * with thr = 0.55, mutual = True): entries > thr that are also their row's and their column's maximum (mutual; ties all count,
 * like the reference's `==`), in nonzero() order.  matches [P, cap, 3] int64 rows (b, i, j), mconf [P, cap] (optional),
 * count [P] = the TRUE number of hits (rows beyond cap are dropped: pass cap = N * M to be safe, min(N, M) + slack in
 * practice), mask [P, N, M] uint8 (optional; the reference's third return value). */
int dr_mutual_match_f64(int P, int N, int M, const double* conf, double thr, int mutual, int cap, int64_t* matches, double* mconf,
                        int32_t* count, uint8_t* mask, void* stream);
int dr_mutual_match_f32(int P, int N, int M, const float* conf, float thr, int mutual, int cap, int64_t* matches, float* mconf,
                        int32_t* count, uint8_t* mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The whole reverse-diffusion loop of Pipeline.forward's eval branch
 * (3D/models/pipeline.py:221-283; 4D/models/pipeline.py:156-197) for P independent pairs, enqueued
 * on `stream` with no host synchronisation (so it can be captured in a HIP graph).
 */
#define DR_VARIANT_3DMATCH 0
#define DR_VARIANT_4DMATCH 1
#define DR_LOOP_STRICT_F64 0x1 /* run the fp64-state Sinkhorn calls in fp64 (streaming kernel) */
/* Pairs of different sizes in one call: pad every pair to (N, M), pass the true extents as masks, and
 * each pair gets exactly the result of its own unpadded B = 1 run: padded rows / columns are excluded from the
 * Sinkhorn marginals (DR_SK_RAGGED), from x.min(), from the top-K rule (K from the true sizes) and from the read-out. */
#define DR_LOOP_RAGGED 0x2
/* The layer GEMMs run on fp16 hi / lo plane images (pgemm, see "Plane images" above) when the configuration allows it
 * (C <= 448, C % 16 == 0) and the call has at least DR_PLANES_MIN_ROWS (environment, default 4096) token rows; these two
 * flags take the decision away from the size rule (tests hold the small golden loops to the reference through both paths) */
#define DR_LOOP_PLANES_FORCE 0x4
#define DR_LOOP_PLANES_OFF 0x8
/* OPT-IN reduced precision (never set by the host mirrors' defaults): on the plane path, the attention's two contractions (q k^T, P v) take ONE
 * fp16 MFMA product of the operands' hi planes instead of the three products that make an fp32-grade one -- what BASELINE's configs[2] / [4]
 * word as "bf16 / fp16 MFMA attention".  Softmax and accumulation stay fp32, the GEMMs keep three products.  11-bit operands: conf_matrix_pred
 * moves by ~1e-3 (bench.py other_configs reports the measured deviation and IR / FMR beside the rate); outside the 1e-4 contract by design. */
#define DR_LOOP_ATTN_F16 0x10

typedef struct {
    int variant;               /* DR_VARIANT_*                                              */
    int C, H, n_layers;        /* feature dim, heads, layers (self, cross, self, ... )       */
    int steps;                 /* SAMPLE_STEP                                                */
    int sk_iters;              /* skh_iters                                                  */
    float voxel, origin[3];    /* voxel_size, vol_bnds[0]                                    */
    float sample_rate, max_condition_num; /* procrustes config                              */
    int flags;                 /* DR_LOOP_*                                                  */
    const double* h_alphas_cumprod; /* HOST [1000] float64 (pipeline.py:151-156)             */
    const int32_t* h_times;    /* HOST [steps+1] reversed int(linspace(0,999,steps+1)), Q20  */
} dr_loop_config;

typedef struct {
    const dr_layer_weights* layers; /* HOST array [n_layers] of device-pointer structs       */
    const float* src_proj;     /* denoising_coarse_matching.src_proj.weight [C,C] (Q1)       */
    const float* bin_score;    /* device pointer to one float                                */
    const float* pe_freq;      /* device [C/6], see dr_vol_pe_f32                            */
    const void* prepacked;     /* dr_loop_prepack image of THESE weights (device, 256-byte aligned) or NULL: the loop
                                * then packs them into its workspace at every call (weights may change between calls) */
} dr_loop_weights;

/* The layer weights as fp16 hi / lo plane images + per-column scales + per-layer LayerNorm bounds, packed ONCE for the
 * life of an engine (immutable weights): pass the buffer as dr_loop_weights.prepacked.  0 bytes = this configuration does
 * not use the plane path (C > 448 or C % 16 != 0): leave prepacked NULL. */
size_t dr_loop_prepack_bytes(const dr_loop_config* cfg);
int dr_loop_prepack(const dr_loop_config* cfg, const dr_loop_weights* w, void* packed, size_t packed_bytes, void* stream);

typedef struct {                /* all optional (NULL to skip); per-step records for parity tests */
    float* x0;                 /* [steps,P,N,M]  x_start of every step                       */
    float* R_forwd;            /* [steps,P,9]                                                */
    float* t_forwd;            /* [steps,P,3]                                                */
    double* cond;              /* [steps,P]                                                  */
    /* the side effects Matching.forward leaves in `data` at the LAST step (3D/models/matching.py:177-187), token layout
     * [P*N + P*M, C] (all src rows, then all tgt rows): src_proj(feats) and its rotary-embedded form (before the 1/sqrt(C)) */
    float* feats_nopos;
    float* feats_pos;
    /* ---- ABI 0.2.0: teacher forcing, for per-step parity tests (tests/test_teacher_forced_gpu.py).  Every step of the loop is then an
     * independent evaluation of one pass through pipeline.py:237-256 (EXP/model.py:637-680) on a state the CALLER supplies, so a top-K
     * near-tie at one step cannot hide the steps behind it.  All optional; a struct zero-filled beyond `feats_pos` behaves like 0.1.0's.
     * force_x   [steps,P,N,M] float64: the state ENTERING step k is force_x[k] instead of the loop's own (step 0: x_T widened; x_T is ignored).
     * force_R / force_t [steps,P,9] / [steps,P,3] (both or neither): the warp of step k is this pose instead of the fit's R_forwd, t_forwd
     *           (the fit still runs and is still traced: R_forwd / t_forwd / cond / topk_idx are the loop's own).
     * x_next    [steps,P,N,M] float64: the state after step k's update (what the loop would carry into step k + 1).
     * topk_idx  [steps,P,K] int32, K = int(max(N,M) * sample_rate): flat indices i * M + j of the entries the fit of step k selected, in arrival
     *           order (a set); slots beyond a pair's own K (mask-length rule) keep -1.
     * wconf     [steps,P,N,M] float32: the warp confidences of step k, float32(exp(Z)[:N,:M]) of pipeline.py:299-302. */
    const double* force_x;
    const float* force_R;
    const float* force_t;
    double* x_next;
    int32_t* topk_idx;
    float* wconf;
} dr_loop_trace;

size_t dr_denoise_loop_workspace_bytes(const dr_loop_config* cfg, int P, int N, int M);

/* inputs : src_feats [P,N,C], tgt_feats [P,M,C], s_pcd [P,N,3], t_pcd [P,M,3] float32;
 *          src_mask [P,N], tgt_mask [P,M] uint8 or NULL; x_T [P,N,M] float32 (the randn of
 *          pipeline.py:224); noise [steps,P,N,M] float32 or NULL (xi, used by the 4D variant only)
 * outputs: conf [P,N,M] float64 (conf_matrix_pred); x_final [P,N,M] float64 (optional);
 *          matches [P,N+M,3] int64 + match_count [P] (3D variant; optional);
 *          R_final [P,9], t_final [P,3] float32: soft_procrustes of float32(conf) (optional; the
 *          reference's own call returns identity through a swallowed dtype error, quirk Q3)
 */
int dr_denoise_loop(const dr_loop_config* cfg, const dr_loop_weights* w, int P, int N, int M,
                    const float* src_feats, const float* tgt_feats, const float* s_pcd, const float* t_pcd,
                    const uint8_t* src_mask, const uint8_t* tgt_mask, const float* x_T, const float* noise,
                    double* conf, double* x_final, int64_t* matches, int32_t* match_count, float* R_final,
                    float* t_final, const dr_loop_trace* trace, void* workspace, size_t workspace_bytes,
                    void* stream);

/* The status of the LAST call that ran on `workspace` (dr_denoise_loop, dr_denoiser_match_f32, dr_denoise_loop_2d3d): waits for `stream`,
 * reads the workspace's own sticky word (zeroed when a call starts; clears it when `clear`) -> DR_OK, or DR_ETIMEOUT when a co-resident
 * Sinkhorn launch OF THAT CALL gave up waiting for another workgroup (its outputs then hold NaN).  Unlike dr_device_status this word
 * belongs to one workspace: concurrent engines (one workspace per stream) cannot swallow or misattribute each other's time-outs. */
int dr_denoise_loop_status(void* workspace, void* stream, int clear);

/* RepositioningTransformer.forward for layer types self/cross (3D/models/transformero.py:151-233)
 * + Matching.forward (3D/models/matching.py:164-219) on already-warped points, for P pairs:
 * one denoiser evaluation.  Outputs: src_out [P,N,C], tgt_out [P,M,C] (optional), conf [P,N,M]. */
int dr_denoiser_match_f32(const dr_loop_config* cfg, const dr_loop_weights* w, int P, int N, int M,
                          const float* src_feats, const float* tgt_feats, const float* s_pcd_warped,
                          const float* t_pcd, const uint8_t* src_mask, const uint8_t* tgt_mask, float* src_out,
                          float* tgt_out, float* conf, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluation harness on device: the consumers of the loop's `match_pred`.
 * Match layout everywhere below: `matches` int64 [P, cap, 3] rows (b, src index, tgt index), the first
 * count[p] rows of segment p valid -- exactly what dr_denoise_loop / dr_top1_union_* write (cap = N+M).
 * The reference's flat [K,3] list of one pair (B = 1, 3D/lib/tester.py:115) is P = 1, cap = K, count = {K};
 * column 0 is not read (the segment is the pair).
 */

/* MatchMotionLoss.compute_inlier_ratio (3D/models/loss.py:383-410): warp the matched source points with the
 * ground-truth pose (+ optional scene flow, the 4DMatch call of 3D/lib/tester.py:267), count matches closer
 * than inlier_thr, float32 arithmetic.  rot [P,9], trn [P,3], s2t_flow [P,N,3] or NULL.
 * out: ir [P] float32 (0 when a pair has fewer than 3 matches, loss.py:403-404), n_inlier [P] int32. */
int dr_inlier_ratio_f32(int P, int cap, int N, int M, const int64_t* matches, const int32_t* count, const float* s_pcd,
                        const float* t_pcd, const float* rot, const float* trn, const float* s2t_flow, float inlier_thr,
                        float* ir, int32_t* n_inlier, void* stream);

/* compute_nrfmr + blend_anchor_motion (3D/lib/tester.py:127-210; knn_point_np 3D/datasets/utils.py:23-40):
 * the matches of a pair are motion anchors (t_pcd[j] - s_pcd[i] at s_pcd[i]); every metric point takes the
 * inverse-distance blend of its 3 nearest anchors (anchors beyond knn_radius get distance 1e10), and counts as
 * recalled when it lands within recall_thr of its ground-truth position rot (p + flow) + trn.
 *   raw_pcd, raw_flow [sum R_p, 3]: the un-subsampled source clouds and their scene flow, pair p at rows
 *   raw_offsets[p] .. raw_offsets[p+1];  metric_index int64 [sum Q_p] (indices into the pair's raw cloud), pair p at
 *   q_offsets[p] .. q_offsets[p+1]  (int32 offsets, P+1 entries each; max_q = the largest Q_p).
 * out: nrfmr [P] float32 (recalled / Q_p; 0 for a pair with fewer than 4 anchors, where the reference's
 * np.argpartition(kth=3) raises); n_recalled [P] int32; blended [sum Q_p, 3] float32 or NULL (the blended motion). */
int dr_nrfmr_f32(int P, int cap, int N, int M, const int64_t* matches, const int32_t* count, const float* s_pcd,
                 const float* t_pcd, const float* raw_pcd, const float* raw_flow, const int32_t* raw_offsets,
                 const int64_t* metric_index, const int32_t* q_offsets, int max_q, const float* rot, const float* trn,
                 float knn_radius, float recall_thr, float* nrfmr, int32_t* n_recalled, float* blended, void* stream);

/* MatchMotionLoss.ransac_regist_coarse -> ransac_pose_estimation (3D/models/loss.py:13-24, 347-379): Open3D 0.13.0
 * (3D/eccv24_3d_env.yml:139) registration_ransac_based_on_correspondence with ransac_n = 3,
 * TransformationEstimationPointToPoint(False), RANSACConvergenceCriteria(50000, 1000) (confidence clamps to 1: no
 * early exit, all `iters` hypotheses are scored).  Hypothesis h of pair p samples three correspondences with
 * replacement from a counter-based generator (splitmix64 of (seed, pair_ids[p], 3h + slot), restated in
 * oracle/metrics_oracle.py -- Open3D's own generator is unseeded, the reference repeats the evaluation three times
 * for that reason, tester.py:24), fits the rigid transform of the triple (Umeyama without scale, fp64), counts the
 * correspondences closer than distance_thr and keeps the best (more inliers; then lower inlier RMSE; then lower h).
 * Triples that repeat a source or a target point (rank-1 covariance: the roll angle is arbitrary) are skipped.
 * out (fp64 like the reference's `torch.from_numpy(pose)`): rot [P,9], trn [P,3]; identity / zero when a pair has
 * fewer than 3 matches (loss.py:363-366) or no hypothesis has an inlier; fitness [P], inlier_rmse [P] (optional),
 * best_iter [P] int32 (optional, -1 = none).   pair_ids int64 [P] or NULL (= 0..P-1). */
size_t dr_ransac_workspace_bytes(int P, int cap, int iters);
int dr_ransac_corr_f64(int P, int cap, int N, int M, const int64_t* matches, const int32_t* count, const float* s_pcd,
                       const float* t_pcd, double distance_thr, int iters, uint64_t seed, const int64_t* pair_ids,
                       double* rot, double* trn, double* fitness, double* inlier_rmse, int32_t* best_iter,
                       void* workspace, size_t workspace_bytes, void* stream);

/* MatchMotionLoss.compute_registration_recall + computeTransformationErr (3D/models/loss.py:27-44, 415-448):
 * e = [t, q_xyz] of inv(gt) * pred (q = unit quaternion of the rotation part, w >= 0; nibabel's mat2quat),
 * err = e^T info e / info[0][0], success = err <= thr^2.  rot_est [P,9], trn_est [P,3] float64; rot_gt [P,9],
 * trn_gt [P,3] float32; info [P,36] float64 (`gt_cov`).  out: err [P] float64, success [P] int32. */
int dr_registration_recall_f64(int P, const double* rot_est, const double* trn_est, const float* rot_gt,
                               const float* trn_gt, const double* info, double thr, double* err, int32_t* success,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Collate-time native code on device: the two C++ extensions the reference's data loader calls to build
 * the KPFCN index arrays (3D/datasets/dataloader.py:13-68, 120-200).  Clouds are stacked: `lengths` int32 [nb] (device).
 * Nothing synchronises; `status` (device int32) becomes 1 when a cloud spans more than 65 533 cells on an axis.
 */

/* cpp_subsampling.subsample_batch(points, batches_len, sampleDl) -> batch_grid_subsampling
 * (cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:4-211, max_p = 0): barycentre of every occupied voxel,
 * summed in float32 in input order (bit-exact).  out_points has room for n rows; the first out_total[0] are written, per
 * cloud in ascending (iz, iy, ix) voxel order (the reference emits std::unordered_map iteration order; consumers are
 * permutation-equivariant); out_lengths [nb]. */
size_t dr_grid_subsample_workspace_bytes(int n, int nb);
int dr_grid_subsample_f32(int n, int nb, const float* points, const int32_t* lengths, float dl, float* out_points,
                          int32_t* out_lengths, int32_t* out_total, int32_t* status, void* workspace, size_t workspace_bytes,
                          void* stream);

/* cpp_neighbors.batch_query(queries, supports, q_batches, s_batches, radius)[:, :limit] -> batch_nanoflann_neighbors
 * (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:210-333) + the truncation of dataloader.py:64-65: for every query the
 * supports of its own cloud with squared distance < radius^2 (float32, nanoflann's L2_Simple_Adaptor order of operations),
 * nearest first (equal distances: lower index first; the reference's std::sort leaves them unspecified), as indices into the
 * stacked supports, padded with ns.  out int64 [nq, limit] (limit <= 64); max_count[0] = the largest neighbourhood found
 * (the reference's matrix is min(max_count, limit) wide: the caller slices). */
size_t dr_radius_neighbors_workspace_bytes(int nq, int ns, int nb);
int dr_radius_neighbors_f32(int nq, int ns, int nb, const float* queries, const float* supports, const int32_t* q_lengths,
                            const int32_t* s_lengths, float radius, int limit, int64_t* out, int32_t* max_count, int32_t* status,
                            void* workspace, size_t workspace_bytes, void* stream);

/* batch_mutual_topk_select / mutual_topk_select (Diff-Reg-2d3d/vision3d/ops/mutual_topk_select.py:7-134; the fine matching
 * behind the 2D-3D loop, EXP/model.py:744-752: k = 2, threshold 0.75, mutual): per batch element the entries among the k best of
 * their row and (mutual) / or of their column, beyond `threshold` when use_threshold (> for largest, < otherwise), inside
 * row_masks [B,N] / col_masks [B,M] (NULL = all).  out_idx int64 [capacity,3] rows (b, i, j) in torch.nonzero order, out_score
 * [capacity]; total[0] = number selected (entries beyond capacity are counted, not written).  k <= min(8, N, M) (DR_EINVAL beyond: torch.topk raises there), N*M <= 262144. */
size_t dr_mutual_topk_workspace_bytes(int B, int N, int M);
int dr_mutual_topk_select_f32(int B, int N, int M, const float* score, int k, int largest, int use_threshold, float threshold,
                              int mutual, const uint8_t* row_masks, const uint8_t* col_masks, int64_t* out_idx, float* out_score,
                              long long capacity, int32_t* total, void* workspace, size_t workspace_bytes, void* stream);

/* The rest of the patch-correspondence block behind the 2D-3D loop (EXP/model.py:707-780), around dr_mutual_topk_select_f32:
 *
 * dr_patch_similarity_f32: for every node correspondence b the similarity of its image patch and its point patch (model.py:726-738):
 *   out[b][i][j] = 0.5 (<img_feats[img_knn_indices[b][i]], pcd_feats[pcd_knn_indices[b][j]]> + 1)
 *   = index_select (vision3d/ops/index_select.py:4-33) of both sides + pairwise_cosine_similarity(normalized = True)
 *   (vision3d/ops/cosine_similarity.py:34-66).  img_feats [*, C], pcd_feats [pcd_rows, C] float32; an index == pcd_rows is the zero row the
 *   reference appends (model.py:707); img_knn_indices [P, Ki], pcd_knn_indices [P, Kc] int64, Kc <= 128; out [P, Ki, Kc].
 * dr_unique_pairs_i64: the duplicate removal of model.py:759-763: sorted distinct values of first[i] * multiplier + second[i] (torch.unique),
 *   unique_keys [n], count[0] = how many.  workspace: dr_unique_pairs_workspace_bytes(n).
 * dr_corr_gather_f32: model.py:761-774 for the first count[0] keys: img / pcd indices = key / num_points_f, key % num_points_f, the gathered
 *   points [.,3] / pixels [.,2] of both sides and corr_scores = <img_feats_f[i], pcd_feats_f[j]>; `capacity` = rows of the outputs. */
int dr_patch_similarity_f32(int P, int Ki, int Kc, int C, const float* img_feats, const int64_t* img_knn_indices, const float* pcd_feats,
                            const int64_t* pcd_knn_indices, long long pcd_rows, float* out, void* stream);
size_t dr_unique_pairs_workspace_bytes(int n);
int dr_unique_pairs_i64(int n, const int64_t* first, const int64_t* second, long long multiplier, int64_t* unique_keys, int32_t* count,
                        void* workspace, size_t workspace_bytes, void* stream);
int dr_corr_gather_f32(int capacity, const int32_t* count, const int64_t* unique_keys, long long num_points_f, int C, const float* img_points_f,
                       const float* img_pixels_f, const float* pcd_points_f, const float* pcd_pixels_f, const float* img_feats_f, const float* pcd_feats_f,
                       int64_t* img_corr_indices, int64_t* pcd_corr_indices, float* img_corr_points, float* img_corr_pixels, float* pcd_corr_points,
                       float* pcd_corr_pixels, float* corr_scores, void* stream);

/* PnP-RANSAC registration of the fine correspondences (EXP/eval.py:174-182 -> vision3d/utils/opencv.py:10-63 = cv2.solvePnPRansac with
 * iterationsCount = 50000, reprojectionError = 8.0, flags = SOLVEPNP_P3P).  OpenCV is not part of the reference tree: the published algorithm of
 * that call -- RANSAC over 4-point samples, P3P on three + disambiguation by the fourth, inliers under the reprojection tolerance, refit on the
 * inliers -- restated in oracle/pnp_oracle.py (every hypothesis is scored; counter-based sampling; Grunert's quartic; Gauss-Newton refit on the
 * reprojection error).  PARITY UNPINNED against OpenCV, pinned against the oracle.
 *   points [n,3], pixels [n,2] float32 device ((h, w) rows when transposed != 0, opencv.py:42-43), intrinsics: 9 doubles on the HOST (row-major K),
 *   transform: 16 doubles (device, row-major 4 x 4, 3D -> camera), n_inlier / best_iter: device ints.  n >= 4. */
size_t dr_pnp_ransac_workspace_bytes(int n, int iters);
int dr_pnp_ransac_f64(int n, const float* points, const float* pixels, int transposed, const double* intrinsics_host, int iters, double distance_tolerance,
                      uint64_t seed, double* transform, int32_t* n_inlier, int32_t* best_iter, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2D-3D variant (Diff-Reg-2d3d): the reverse sampling of MATR2D3D.forward
 * (EXP/model.py:637-694, 830-846; EXP = Diff-Reg-2d3d/experiments/2d3dmatr.rgbdv2.stage4.level3.stage1)
 * with CrossModalFusionModule (EXP/fusion_module.py:61-107, vision3d/layers/transformer.py:58-301)
 * as the denoiser and the position-free Matching head (EXP/matching.py:91-147).
 * src = N point-cloud nodes, tgt = M image patches.
 */
typedef struct {        /* one vision3d TransformerLayer: weights [out,in] row-major, biases [out] */
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b;     /* attention.attention.{q,k,v}_token_layer   */
    const float *lin_w, *lin_b, *norm1_w, *norm1_b;     /* attention.linear, attention.norm          */
    const float *expand_w, *expand_b, *squeeze_w, *squeeze_b, *norm2_w, *norm2_b; /* output.*        */
} dr_fusion_layer_weights;

typedef struct {
    const dr_fusion_layer_weights* layers;   /* HOST array [n_layers] */
    const float *img_emb_w, *img_emb_b;      /* img_emb_proj: weight zero-padded to [C, 44] (K = 42 -> 44)  */
    const float *pcd_emb_w, *pcd_emb_b;      /* pcd_emb_proj: weight zero-padded to [C, 64] (K = 63 -> 64)  */
    const float *img_in_w, *img_in_b;        /* img_in_proj      [C, img_dim]                               */
    const float *dino_w, *dino_b;            /* img_in_proj_dino [C, dino_dim]                              */
    const float *all_w, *all_b;              /* img_in_proj_all  [C, 2C]                                    */
    const float *pcd_in_w, *pcd_in_b;        /* pcd_in_proj      [C, pcd_dim]                               */
    const float *out_w, *out_b;              /* out_proj         [C, C]                                     */
    const float* src_proj;                   /* denoising_coarse_matching.src_proj.weight [C,C] (Q1)        */
    const float* bin_score;
    const void* prepacked;                   /* dr_loop2d3d_prepack image of THESE weights (device, 256-byte aligned) or NULL: calls that
                                              * take the plane path then pack them into the workspace EVERY time -- 7 pack launches, 3 copies
                                              * and 6 bound kernels per layer, also inside a captured graph (every replay repeats them): pass a
                                              * prepacked image unless the weights really change between calls.  The image must be complete on
                                              * the stream of the call: synchronise (or order with an event) after dr_loop2d3d_prepack when the
                                              * loop runs on another stream                                                            */
} dr_fusion_weights;

typedef struct {
    int C, H, n_layers;          /* hidden/output dim (256), heads (4), blocks (self, cross, ...)   */
    int img_dim, dino_dim, pcd_dim;
    int steps, sk_iters;
    float sample_rate, max_condition_num;
    int flags;                   /* DR_LOOP_*                                                     */
    const double* h_alphas_cumprod;
    const int32_t* h_times;
} dr_loop2d3d_config;

/* Calls of at least 4096 token rows (P (N + M); cfg5: two pairs or more) run the layer's nn.Linears and its attention on fp16 hi / lo
 * plane images like dr_denoise_loop does ("Plane images" above; biases, the post-add LayerNorms and the ReLU feed-forward of the vision3d
 * layer are epilogue modes of the same GEMM); DR_LOOP_PLANES_FORCE / DR_LOOP_PLANES_OFF in cfg->flags take the decision away from the size
 * rule.  The packed weights: once per engine with dr_loop2d3d_prepack (0 bytes = this configuration has no plane path). */
size_t dr_loop2d3d_prepack_bytes(const dr_loop2d3d_config* cfg);
int dr_loop2d3d_prepack(const dr_loop2d3d_config* cfg, const dr_fusion_weights* w, void* packed, size_t packed_bytes, void* stream);
size_t dr_denoise_loop_2d3d_workspace_bytes(const dr_loop2d3d_config* cfg, int P, int N, int M);

/* inputs : img_feats [P,M,img_dim], img_dino [P,M,dino_dim], img_pixels [P,M,2], pcd_feats [P,N,pcd_dim],
 *          s_pcd [P,N,3] (point nodes), t_pcd_da [P,M,3] (depth-back-projected patch centres),
 *          src_mask [P,N], tgt_mask [P,M], tgt_mask_da [P,M] uint8 (all three or none), x_T [P,N,M]
 * outputs: conf [P,N,M] float64; matches [P,N+M,3] int64 + match_count [P] (optional); x_final (optional);
 *          trace as in dr_denoise_loop; steps == 0 runs ONE fusion + matching evaluation on s_pcd as given
 *          and writes x_start into conf as float64 (used by the component tests) */
int dr_denoise_loop_2d3d(const dr_loop2d3d_config* cfg, const dr_fusion_weights* w, int P, int N, int M,
                         const float* img_feats, const float* img_dino, const float* img_pixels,
                         const float* pcd_feats, const float* s_pcd, const float* t_pcd_da, const uint8_t* src_mask,
                         const uint8_t* tgt_mask, const uint8_t* tgt_mask_da, const float* x_T, double* conf,
                         double* x_final, int64_t* matches, int32_t* match_count, float* img_out, float* pcd_out,
                         const dr_loop_trace* trace, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Forward half of the training branch: the pieces of Pipeline.forward's
 * `if self.training:` block (3D/models/pipeline.py:182-216) and of MatchMotionLoss.ge_coarse_loss
 * (3D/models/loss.py:80-170) that are not already kernels of the loop.  Values only: no backward.
 * `workspace`: dr_train_workspace_bytes(P, N, M) bytes, any content.
 */
#define DR_MATCH_SINKHORN 0
#define DR_MATCH_DUAL_SOFTMAX 1
size_t dr_train_workspace_bytes(int P, int N, int M);

/* match_2_conf_matrix (loss.py:316-320) and matrix_gt (pipeline.py:203-206): out [P,N,M] = 0, then 1 at the rows (b, i, j) of
 * matches [K,3] int64 (duplicates set the same entry; rows outside the matrix are skipped -- the reference raises) */
int dr_match_matrix_f32(int P, int N, int M, int K, const int64_t* matches, float* out, void* stream);

/* pipeline.py:209-214: noise = (|r| % 1) * (|r| / r) * 1.5 of randn [P,N,M] (float32 arithmetic), q_sample(x_start = matrix_gt, t,
 * noise) = sqrt(ac_t) matrix_gt + sqrt(1 - ac_t) noise in float64 (pipeline.py:84-95: the schedule is float64 and promotes),
 * nan_to_num(nan = 0), minus the minimum over the WHOLE [P,N,M] tensor (not per pair).  Bit-exact against the reference. */
int dr_gt_noising_f64(int P, int N, int M, const float* matrix_gt, const float* randn, double sqrt_ac, double sqrt_one_minus_ac, double* out,
                      void* workspace, void* stream);

/* MatchMotionLoss.compute_correspondence_loss (loss.py:273-314): focal loss of conf [P,N,M] against conf_gt (entries 1 / 0) ->
 * *loss (device float).  DR_MATCH_SINKHORN: pos_w mean_pos(-alpha (1-c)^gamma log c) + neg_w mean_neg(-alpha c^gamma log(1-c));
 * DR_MATCH_DUAL_SOFTMAX: pos_w mean_pos(... weight).  c clamped to [1e-6, 1-1e-6]; a class without entries contributes 0 (the
 * reference's stand-in entry with weight 0).  Terms in float32 in the reference's operation order, summed in float64. */
int dr_focal_loss_f32(int P, int N, int M, const float* conf, const float* conf_gt, const float* weight, float alpha, float gamma, float pos_w,
                      float neg_w, int match_type, float* loss, void* workspace, void* stream);

/* MatchMotionLoss.compute_match_recall (loss.py:323-345): match_pred [K,3] int64 rows (b, i, j) -> recall_precision[0] = TP / sum(conf_gt),
 * [1] = TP / max(K, 1), TP = ground-truth entries among the (distinct) predicted ones */
int dr_match_recall_f32(int P, int N, int M, const float* conf_gt, int K, const int64_t* match_pred, float* recall_precision, void* workspace,
                        void* stream);

/* the L1 motion term (loss.py:108-128): mean over the rows with overlap_mask [P,N] set of |(R_pred s + t_pred) - (R_gt (s + flow) + t_gt)|_1;
 * s_pcd [P,N,3], flow [P,N,3] or NULL (3DMatch), R [P,3,3], t [P,3] float32 */
int dr_motion_l1_f32(int P, int N, const float* s_pcd, const float* flow, const float* R_pred, const float* t_pred, const float* R_gt,
                     const float* t_gt, const uint8_t* overlap_mask, float* loss, void* workspace, void* stream);

/* First backward kernels of the training branch: the matching head's loss.
 * dr_focal_loss_backward_f32: d loss / d conf of compute_correspondence_loss in its sinkhorn form (loss.py:311-314; torch.clamp passes no
 *   gradient outside [1e-6, 1 - 1e-6]); workspace: dr_train_workspace_bytes(P, N, M).
 * dr_sinkhorn_backward_f32: backward of log_optimal_transport + exp + [:, :-1, :-1] (matching.py:61-93, 207-216): scores [P,N,M] as the forward
 *   saw them (masked entries -inf), the masks (both or none), bin_score, iters, grad_conf = d loss / d conf [P,N,M] ->
 *   grad_scores [P,N,M] (0 at masked entries) and grad_bin_score [P] (one partial per pair: the caller sums them).  float32 in and out; the
 *   dual variables, the plans' exponents and every sum are kept in double (at logits in the thousands Z + u + v cancels: a float32
 *   exponent cost 2.7e-3 of the gradient's maximum); every reduction in a fixed order.  workspace: dr_sinkhorn_backward_workspace_bytes(P, N, M, iters). */
/* d loss / d (R_pred [P,3,3], t_pred [P,3]) of dr_motion_l1_f32 (loss.py:108-128; 4DMatch trains with motion_weight 0.1) */
int dr_motion_l1_backward_f32(int P, int N, const float* s_pcd, const float* flow, const float* R_pred, const float* t_pred, const float* R_gt,
                              const float* t_gt, const uint8_t* overlap_mask, float* grad_R, float* grad_t, void* workspace, void* stream);

/* The non-GEMM pieces of the GeometryAttentionLayer backward (transformero.py:43-96; composed with dr_linear_f32 by diffreg_hip/autograd.py):
 * nn.LayerNorm forward with saved statistics (mean_rstd [rows,2]) and its backward (grad_x, grad_gamma, grad_beta; workspace:
 * dr_layernorm_backward_workspace_bytes(C)); the masked softmax of an explicit attention matrix scores [B,H,L,S] over S (a key j is -inf
 * for a valid query: q_mask[b][l] && !k_mask[b][j], transformero.py:78-79; then / sqrt(d) = `scale`) and its backward
 * dS = scale P (dP - sum_j dP_j P_j); ReLU backward. */
int dr_layernorm_f32(int rows, int C, const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean_rstd, void* stream);
size_t dr_layernorm_backward_workspace_bytes(int C);
int dr_layernorm_backward_f32(int rows, int C, const float* x, const float* gamma, const float* mean_rstd, const float* grad_y, float* grad_x,
                              float* grad_gamma, float* grad_beta, void* workspace, void* stream);
/* softmax(q k^T scale) v per head, fused (no [B,H,L,S] matrix): forward on the inference kernels, and its backward (3D/models/transformero.py:79-85
 * under autograd).  Token layout: q / out / grad_o [B L, ld], k / v [B S, ld], head h in columns [h d, (h + 1) d); masks [B L] / [B S] uint8 (both
 * or none): key j is dead for query l when q_mask[l] && !k_mask[j], as the training forward applies them.  The backward is three launches on the
 * f32-input MFMA (per-query log-sum-exp and delta; dQ by query blocks; dK | dV by key blocks), fixed summation orders (bit-reproducible);
 * workspace: dr_attention_backward_workspace_bytes(B, H, L).  d % 4 == 0, d <= 160. */
int dr_attention_f32(int B, int H, int L, int S, int d, const float* q, const float* k, const float* v, int ld, const uint8_t* q_mask,
                     const uint8_t* k_mask, float scale, float* out, void* stream);
size_t dr_attention_backward_workspace_bytes(int B, int H, int L);
int dr_attention_backward_f32(int B, int H, int L, int S, int d, const float* q, const float* k, const float* v, const float* o, const float* grad_o,
                              int ld, const uint8_t* q_mask, const uint8_t* k_mask, float scale, float* grad_q, float* grad_k, float* grad_v,
                              void* workspace, size_t workspace_bytes, void* stream);
int dr_softmax_rows_f32(int B, int H, int L, int S, const float* scores, float scale, const uint8_t* q_mask, const uint8_t* k_mask, float* P, void* stream);
int dr_softmax_backward_f32(int rows, int cols, const float* P, const float* grad_P, float scale, float* grad_scores, void* stream);
/* The dual-softmax read-out of Matching.forward (3D/models/matching.py:193-205; match_type 'dual_softmax'):
 *   conf[p][i][j] = softmax_i(sim[p][.][j] / temperature over the valid source rows) * softmax_j(sim[p][i][.] / temperature over the valid target columns),
 *   0 where either mask is 0; masks uint8 [P,N] / [P,M] or both NULL (the reference's mask-free branch).  col_stats: caller scratch of 2 P M floats. */
int dr_dual_softmax_f32(int P, int N, int M, const float* sim, float temperature, const uint8_t* src_mask, const uint8_t* tgt_mask, float* conf,
                        float* col_stats, void* stream);
/* its backward (ABI 0.2.1): grad_sim = d loss / d sim given grad_conf = d loss / d conf -- the two softmax adjoints; entries under a mask receive 0, as
 * the reference's masked_fill_ lets nothing through.  workspace: dr_dual_softmax_backward_workspace_bytes (column / row statistics and sums). */
size_t dr_dual_softmax_backward_workspace_bytes(int P, int N, int M);
int dr_dual_softmax_backward_f32(int P, int N, int M, const float* sim, float temperature, const uint8_t* src_mask, const uint8_t* tgt_mask,
                                 const float* grad_conf, float* grad_sim, void* workspace, size_t workspace_bytes, void* stream);
int dr_relu_backward_f32(long long n, const float* y, const float* grad_y, float* grad_x, void* stream);

/* embed_rotary (position_encoding.py:25-35) on contiguous rows [rows, C]: out = R(theta) x * scale with cos / sin [rows, C/2]; inverse != 0:
 * R(-theta), the transpose = the backward of the embedding */
int dr_rotary_f32(int rows, int C, const float* x, const float* cos_t, const float* sin_t, int inverse, float scale, float* out, void* stream);
int dr_focal_loss_backward_f32(int P, int N, int M, const float* conf, const float* conf_gt, float alpha, float gamma, float pos_w, float neg_w,
                               float* grad_conf, void* workspace, void* stream);
size_t dr_sinkhorn_backward_workspace_bytes(int P, int N, int M, int iters);
int dr_sinkhorn_backward_f32(int P, int N, int M, const float* scores, const uint8_t* src_mask, const uint8_t* tgt_mask, const float* bin_score,
                             int iters, const float* grad_conf, float* grad_scores, float* grad_bin_score, void* workspace,
                             size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFREG_HIP_H */
