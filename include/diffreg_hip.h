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

int dr_version(void);                 /* major*10000 + minor*100 + patch */
const char* dr_strerror(int code);
const char* dr_last_hip_error(void);  /* text of the last failing HIP call on this thread */

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

size_t dr_sinkhorn_workspace_bytes(int B, int N, int M, int elem_bytes, int flags);

int dr_sinkhorn_f32(int B, int N, int M, const float* scores, const uint8_t* src_mask,
                    const uint8_t* tgt_mask, const float* bin_score, int iters, int flags,
                    float* out, void* workspace, size_t workspace_bytes, void* stream);

int dr_sinkhorn_f64(int B, int N, int M, const double* scores, const uint8_t* src_mask,
                    const uint8_t* tgt_mask, const float* bin_score, int iters, int flags,
                    void* out /* double*, or float* with DR_SK_OUT_F32 */, void* workspace,
                    size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFREG_HIP_H */
