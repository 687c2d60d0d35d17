// calibration: fp32-input MFMA issue rate with 1 / 2 / 4 independent accumulator chains, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0; for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH> void run(int blocks, int threads) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 8 * CH * 4096.0;
    printf("chains %d blocks %d threads %d : %.3f ms  %.1f TFLOP/s\n", CH, blocks, threads, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, 256); run<2>(256, 256); run<4>(256, 256);
    run<1>(512, 256); run<2>(512, 256); run<1>(256, 512); run<1>(1024, 256);
    return 0;
}
