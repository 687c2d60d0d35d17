// where.hip -- tool (not in the product): which XCC / SE / CU a workgroup runs on, for probing CU-masked streams.
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/where.hip -o tools/_build/libwhere.so
#include <hip/hip_runtime.h>
__global__ void where_kernel(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
extern "C" int where_launch(unsigned* out, int nwg, int spin, void* stream) {
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(64), 0, (hipStream_t)stream, out, spin);
    return (int)hipGetLastError();
}
extern "C" int masked_stream(void** st, int nwords, const unsigned* mask) {
    return (int)hipExtStreamCreateWithCUMask((hipStream_t*)st, (unsigned)nwords, mask);
}
