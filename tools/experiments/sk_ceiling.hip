// Copy-rate ceilings in the Sinkhorn tile access pattern on this box, beside dr_sinkhorn_f32 on the same 4096 tiles (a tool,
// not part of the product).  Build + run:
//   hipcc -O3 --offload-arch=gfx950 tools/sk_ceiling.hip -o tools/_build/sk_ceiling -Ldiff-reg_amd/diffreg_hip -ldiffreg_hip \
//         -Wl,-rpath,'$ORIGIN/../../diff-reg_amd/diffreg_hip' && ./tools/_build/sk_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/diffreg_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_f4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// each thread: U float4 loads in flight, then U stores
template <int U>
__global__ __launch_bounds__(256) void copy_f4_u(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    for (; base + 256ul * (U - 1) < n; base += (size_t)gridDim.x * 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[base + 256ul * u];
#pragma unroll
        for (int u = 0; u < U; ++u) out[base + 256ul * u] = v[u];
    }
}
// tile pattern: one 1024-thread workgroup per 256x256 tile, 16 float4 per thread (the sk_fast layout)
__global__ __launch_bounds__(1024) void copy_tile(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* src = in + (size_t)blockIdx.x * 65536 + w * 16 * 256 + lane * 4;
    float* dst = out + (size_t)blockIdx.x * 65536 + w * 16 * 256 + lane * 4;
    float4 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = *(const float4*)(src + r * 256);
#pragma unroll
    for (int r = 0; r < 16; ++r) *(float4*)(dst + r * 256) = v[r];
}
template <bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_tile_nt(const float* __restrict__ in, float* __restrict__ out) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* src = in + (size_t)blockIdx.x * 65536 + w * 16 * 256 + lane * 4;
    float* dst = out + (size_t)blockIdx.x * 65536 + w * 16 * 256 + lane * 4;
    v4f v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = NTL ? __builtin_nontemporal_load((const v4f*)(src + r * 256)) : *(const v4f*)(src + r * 256);
#pragma unroll
    for (int r = 0; r < 16; ++r) { if (NTS) __builtin_nontemporal_store(v[r], (v4f*)(dst + r * 256)); else *(v4f*)(dst + r * 256) = v[r]; }
}
// same, with a dummy dependent compute phase of `spin` FMA rounds between load and store
__global__ __launch_bounds__(1024) void copy_tile_spin(const float* __restrict__ in, float* __restrict__ out, int spin) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* src = in + (size_t)blockIdx.x * 65536 + w * 16 * 256 + lane * 4;
    float* dst = out + (size_t)blockIdx.x * 65536 + w * 16 * 256 + lane * 4;
    float4 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = *(const float4*)(src + r * 256);
    for (int s = 0; s < spin; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[r].x = fmaf(v[r].x, 1.0001f, 0.5f); v[r].y = fmaf(v[r].y, 1.0001f, 0.5f); v[r].z = fmaf(v[r].z, 1.0001f, 0.5f); v[r].w = fmaf(v[r].w, 1.0001f, 0.5f); }
#pragma unroll
    for (int r = 0; r < 16; ++r) *(float4*)(dst + r * 256) = v[r];
}

template <typename F>
static double time_us(F&& f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int B = 4096;
    const size_t n = (size_t)B * 65536;
    float *in, *out, *alpha;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&alpha, 4));
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f * 8.f - 4.f; }
    CK(hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice));
    float one = 1.f; CK(hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice));
    const double bytes = 2.0 * n * 4;
    auto rep = [&](const char* name, double us) { printf("%-34s %9.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000); };
    dr_init();
    for (int g : {2048, 4096, 8192, 16384})
        { char nm[64]; snprintf(nm, 64, "copy_f4 grid %d", g); rep(nm, time_us([&] { copy_f4<<<g, 256>>>((const float4*)in, (float4*)out, n / 4); })); }
    for (int g : {1024, 2048, 4096, 8192})
        { char nm[64]; snprintf(nm, 64, "copy_f4_u<4> grid %d", g); rep(nm, time_us([&] { copy_f4_u<4><<<g, 256>>>((const float4*)in, (float4*)out, n / 4); })); }
    for (int g : {1024, 2048, 4096})
        { char nm[64]; snprintf(nm, 64, "copy_f4_u<8> grid %d", g); rep(nm, time_us([&] { copy_f4_u<8><<<g, 256>>>((const float4*)in, (float4*)out, n / 4); })); }
    rep("copy_tile (1 WG = 1 tile)", time_us([&] { copy_tile<<<B, 1024>>>(in, out); }));
    rep("copy_tile nt loads", time_us([&] { copy_tile_nt<true, false><<<B, 1024>>>(in, out); }));
    rep("copy_tile nt stores", time_us([&] { copy_tile_nt<false, true><<<B, 1024>>>(in, out); }));
    rep("copy_tile nt both", time_us([&] { copy_tile_nt<true, true><<<B, 1024>>>(in, out); }));
    rep("copy_tile (again)", time_us([&] { copy_tile<<<B, 1024>>>(in, out); }));
    for (int sp : {64})
        { char nm[64]; snprintf(nm, 64, "copy_tile_spin %d (x64 fma/thr)", sp); rep(nm, time_us([&] { copy_tile_spin<<<B, 1024>>>(in, out, sp); })); }
    for (int it : {0, 1, 3})
        { char nm[64]; snprintf(nm, 64, "dr_sinkhorn_f32 iters=%d", it); rep(nm, time_us([&] { dr_sinkhorn_f32(B, 256, 256, in, nullptr, nullptr, alpha, it, 0, out, nullptr, 0, nullptr); })); }
    return 0;
}
