// What the chip sustains on dense fp16 MFMAs under its own clock management: bare loops on RANDOM operands (zeros run at 2.4 GHz and hide it), operands
// in registers, 28 accumulator tiles of 16 x 16 (or 7 of 32 x 32) per wave like the plane GEMM, 1 or 2 waves per SIMD on every CU, >= 0.5 s of back-to-back
// launches before the timed ones.  Prints TFLOP/s of fp16 MFMA work (the nominal dense peak is 2 516.6) and the in-kernel clock.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_ceiling.hip -o tools/_build/mfma_f16_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ in, float* out, int iters, unsigned long long* clk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[7];
    for (int i = 0; i < 4; ++i) a[i] = in[(t * 11 + i) & 65535];
    for (int j = 0; j < 7; ++j) b[j] = in[(t * 11 + 4 + j) & 65535];
    f32x4 c16[4][7];
    f32x16 c32[7];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 7; ++j) c16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 7; ++j) for (int r = 0; r < 16; ++r) c32[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 16) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 7; ++j) c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], c16[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int j = 0; j < 7; ++j) c32[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rep & 3], b[j], c32[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 7; ++j) s += c16[i][j][0] + c16[i][j][3];
    for (int j = 0; j < 7; ++j) s += c32[j][0] + c32[j][15];
    out[t] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int SHAPE> void run(int threads, const f16x8* in) {
    const int blocks = 256;
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&clk, blocks * 16);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, clk);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    unsigned long long h[512]; hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
    double ghz = 0; for (int i = 0; i < blocks; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; ghz /= blocks;
    const double per_iter = SHAPE == 16 ? 84 * 16384.0 : 42 * 32768.0;   // MFMA flops of a wave per iteration
    const double flops = (double)blocks * (threads / 64) * iters * per_iter;
    printf("v_mfma_f32_%s_f16, %d waves per SIMD: %.3f ms  %.1f TFLOP/s = %.3f of 2516.6; in-kernel clock %.2f GHz\n", SHAPE == 16 ? "16x16x32" : "32x32x16",
           threads / 256, ms, flops / ms / 1e9, flops / ms / 1e9 / 2516.6, ghz);
    hipFree(out); hipFree(clk);
}
int main() {
    f16x8* in; hipMalloc(&in, 65536 * sizeof(f16x8));
    _Float16* h = (_Float16*)malloc(65536 * 16);
    srand(7);
    for (int i = 0; i < 65536 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 200.f);
    hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    run<32>(256, in); run<32>(512, in); run<16>(256, in); run<16>(512, in);
    return 0;
}
