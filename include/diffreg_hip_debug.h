/* diffreg_hip_debug.h -- diagnostics of libdiffreg_hip.so for tools/ and for the kernel-forcing test fixtures.
 * NOT part of the drop-in boundary (include/diffreg_hip.h): nothing a deployment calls is declared here.
 *
 * The library reads no environment variable unless dr_debug_enable_env(1) was called: the DR_* tuning variables of tools/
 * (DR_GEMM_*, DR_ATTN_*, DR_PLANES*, DR_PG_*, DR_SK_PERSIST_GRID) cannot change which kernels a deployment runs.  Once enabled they are
 * re-read on every launch (nothing is latched).  The plane GEMM's: DR_PG_HALF (0 / 2: never / always 64-row workgroups), DR_PG_HALF_PCT (the
 * 64-row rule's threshold in percent of the CU count), DR_PG_M16 (0: the 32x32x16 main loop instead of the 16x16x32 one), DR_PG_NOEPI (timing
 * ablations, WRONG results: bit 0 return behind the main loop, 2 no fp32 row stores, 3 no image stores, 4 no residual loads, 5 no rotary tables).  Per-call choices of the product path are arguments: dr_loop_config.flags
 * (DR_LOOP_PLANES_FORCE / _OFF, DR_LOOP_STRICT_F64, DR_LOOP_RAGGED), DR_SK_* flags.  The setters below are process-wide and meant
 * for single-threaded tools and tests only. */
#ifndef DIFFREG_HIP_DEBUG_H
#define DIFFREG_HIP_DEBUG_H
#include "diffreg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

void dr_debug_enable_env(int on);
/* force the f32-input GEMM configuration: -1 auto, 0 / 1 / 2 / 9 LDS-staged 32 x 32 / 64 x 64 / 128 x 64 / 64 x 64 single-buffer
 * tiles, 11 / 12 the latency form with 8 / 16 waves */
void dr_debug_gemm_config(int c);
/* attention: use the 128-query (flash) kernel from this many workgroups on; -1 = default rule (256) */
void dr_debug_attention_config(int flash_min_workgroups);
/* flash attention arithmetic: 1 = split-operand bf16 MFMA products (default), 0 = f32-input MFMA, -1 = default */
void dr_debug_attention_split(int on);
/* polls before a workgroup of the single-launch Sinkhorn gives up waiting for the others (0 = the default, 2^22); the timeout test
 * sets 1 so that the first failed poll already gives up (tests/test_errors_gpu.py) */
void dr_debug_sinkhorn_spin_limit(unsigned polls);
/* n dependent launches of an empty kernel (tools/launch_floor.py) */
int dr_debug_launch_chain(int n, int workgroups, int threads, void* stream);

#ifdef __cplusplus
}
#endif
#endif
