// common.h -- shared device/host helpers for libdiffreg_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/diffreg_hip_debug.h"

namespace dr {

constexpr int WAVE = 64;

void set_hip_error(hipError_t e, const char* where);
// Tuning / diagnostic overrides from the environment (tools/): read ONLY after dr_debug_enable_env(1) -- the product path never
// enables it, so a stray DR_* variable in a deployment's environment cannot change which kernels run.
int env_knob(const char* name, int def);
void enable_env_knobs(bool on);
// compute units of the CURRENT device (cached per device index: a process may drive devices with different CU counts)
int device_cu_count();

#define DR_HIP_CHECK(expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) {                              \
            dr::set_hip_error(_e, #expr);                    \
            return DR_ELAUNCH;                               \
        }                                                    \
    } while (0)

#define DR_LAUNCH_CHECK()                                    \
    do {                                                     \
        hipError_t _e = hipGetLastError();                   \
        if (_e != hipSuccess) {                              \
            dr::set_hip_error(_e, "kernel launch");          \
            return DR_ELAUNCH;                               \
        }                                                    \
    } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        T o = __shfl_xor(v, m);
        v = v > o ? v : o;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        T o = __shfl_xor(v, m);
        v = v < o ? v : o;
    }
    return v;
}

}  // namespace dr
