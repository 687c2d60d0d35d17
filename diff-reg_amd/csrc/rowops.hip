// rowops.hip -- HBM-bound row-wise kernels: LayerNorm(+residual), volumetric rotary tables.
#include "kernels.h"

namespace dr {

// ---------------------------------------------------------------------------------------------
// out[r] = res[r] + LN(x[r]) * g + b   -- one wave per row, row held in registers (C <= 1024)
// nn.LayerNorm semantics (biased variance, eps inside the sqrt), transformero.py:40-41,88-94
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAX_PER_LANE = 16;   // C <= 64 * 16

template <bool POSTADD>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                        const float* __restrict__ b, const float* __restrict__ res,
                                                        int ldres, float* __restrict__ out, int ldo, int rows, int C,
                                                        float* __restrict__ rowmax) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * ldx;
    float ymax = 0.f;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? xr[c] : 0.f;
        if (POSTADD && c < C) v[i] += res[(size_t)row * ldres + c];
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            float y = (v[i] - mean) * rstd * g[c] + b[c];
            if (!POSTADD && res) y += res[(size_t)row * ldres + c];
            out[(size_t)row * ldo + c] = y;
            ymax = fmaxf(ymax, fabsf(y));
        }
    }
    if (rowmax) { ymax = wave_max(ymax); if (lane == 0) rowmax[row] = ymax; }
}

// float4 form for C % 4 == 0 with 16-byte aligned rows (every layer of the loop: C = 432 / 528 / 256): a row is C / 4 float4s,
// one wave per row, NV = ceil(C / 256) 16-byte loads per lane instead of 16 predicated dword loads
template <bool POSTADD, int NV>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                            const float* __restrict__ b, const float* __restrict__ res,
                                                            int ldres, float* __restrict__ out, int ldo, int rows, int C,
                                                            float* __restrict__ rowmax) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, C4 = C >> 2;
    float ymax = 0.f;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
    const float4* rr = res ? reinterpret_cast<const float4*>(res + (size_t)row * ldres) : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (POSTADD && c < C4) { const float4 r = rr[c]; v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w; }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < C4) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < C4) {
            const float4 gg = reinterpret_cast<const float4*>(g)[c], bb = reinterpret_cast<const float4*>(b)[c];
            float4 y;
            y.x = (v[i].x - mean) * rstd * gg.x + bb.x; y.y = (v[i].y - mean) * rstd * gg.y + bb.y;
            y.z = (v[i].z - mean) * rstd * gg.z + bb.z; y.w = (v[i].w - mean) * rstd * gg.w + bb.w;
            if (!POSTADD && rr) { const float4 r = rr[c]; y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w; }
            reinterpret_cast<float4*>(out + (size_t)row * ldo)[c] = y;
            ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w))));
        }
    }
    if (rowmax) { ymax = wave_max(ymax); if (lane == 0) rowmax[row] = ymax; }
}

template <bool POSTADD>
static bool launch_layernorm_vec(const float* x, int ldx, const float* g, const float* b, const float* res, int ldres, float* out,
                                 int ldo, int rows, int C, hipStream_t st, float* rowmax) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if ((C & 3) || (ldx & 3) || (ldo & 3) || (res && (ldres & 3)) || !al(x) || !al(out) || !al(g) || !al(b) || (res && !al(res)) || C > 768)
        return false;
    const dim3 grid((rows + 3) / 4), blk(256);
    if (C <= 256) hipLaunchKernelGGL((layernorm_vec_kernel<POSTADD, 1>), grid, blk, 0, st, x, ldx, g, b, res, ldres, out, ldo, rows, C, rowmax);
    else if (C <= 512) hipLaunchKernelGGL((layernorm_vec_kernel<POSTADD, 2>), grid, blk, 0, st, x, ldx, g, b, res, ldres, out, ldo, rows, C, rowmax);
    else hipLaunchKernelGGL((layernorm_vec_kernel<POSTADD, 3>), grid, blk, 0, st, x, ldx, g, b, res, ldres, out, ldo, rows, C, rowmax);
    return true;
}

int launch_layernorm(const float* x, int ldx, const float* g, const float* b, const float* res, int ldres, float* out,
                     int ldo, int rows, int C, hipStream_t st, float* rowmax) {
    if (C > 64 * LN_MAX_PER_LANE) return DR_ENOSUP;
    if (rows <= 0) return DR_OK;
    ProfScope ps(PK_LN, (double)rows * C * (res ? 12.0 : 8.0), st);
    if (launch_layernorm_vec<false>(x, ldx, g, b, res, ldres, out, ldo, rows, C, st, rowmax)) { DR_LAUNCH_CHECK(); return DR_OK; }
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, g, b, res, ldres, out, ldo, rows, C, rowmax);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_layernorm_postadd(const float* x, int ldx, const float* g, const float* b, const float* res, int ldres, float* out,
                             int ldo, int rows, int C, hipStream_t st) {
    if (C > 64 * LN_MAX_PER_LANE || !res) return DR_ENOSUP;
    if (rows <= 0) return DR_OK;
    ProfScope ps(PK_LN, (double)rows * C * 12.0, st);
    if (launch_layernorm_vec<true>(x, ldx, g, b, res, ldres, out, ldo, rows, C, st, nullptr)) { DR_LAUNCH_CHECK(); return DR_OK; }
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, g, b, res, ldres, out, ldo, rows, C, (float*)nullptr);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---------------------------------------------------------------------------------------------
// FourierEmbedding(length L, use_pi=False, use_input=True) (2D3D/vision3d/layers/embedding.py:75-100):
// [p (D) | level 0: sin (D), cos (D) | level 1: ... ], factors 2^l exact in float32
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_mean_kernel(const float* __restrict__ xyz, int n, const float* __restrict__ R,
                                                        const float* __restrict__ tv, float* __restrict__ warped) {
    // one workgroup per pair: warped = R p + t, then subtract the mean over the n rows (points.mean(dim=1))
    __shared__ double s[4][3];
    const int pair = blockIdx.x, t = threadIdx.x;
    const float* p = xyz + (size_t)pair * n * 3;
    float* wp = warped + (size_t)pair * n * 3;
    double acc[3] = {0, 0, 0};
    for (int i = t; i < n; i += 256) {
        float q[3] = {p[i * 3], p[i * 3 + 1], p[i * 3 + 2]};
        if (R) {
            const float* Rp = R + pair * 9;
            const float* tp = tv + pair * 3;
            float o[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
                o[a] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Rp[a * 3], q[0]), __fmul_rn(Rp[a * 3 + 1], q[1])), __fmul_rn(Rp[a * 3 + 2], q[2])), tp[a]);
            q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { wp[i * 3 + a] = q[a]; acc[a] += (double)q[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v = wave_sum(acc[a]);
        if ((t & 63) == 0) s[t >> 6][a] = v;
    }
    __syncthreads();
    float mean[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) mean[a] = (float)((s[0][a] + s[1][a] + s[2][a] + s[3][a]) / (double)n);
    __syncthreads();
    for (int i = t; i < n; i += 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) wp[i * 3 + a] = __fsub_rn(wp[i * 3 + a], mean[a]);
}

__global__ __launch_bounds__(256) void fourier_kernel(const float* __restrict__ p, int rows, int D, int L, float* __restrict__ emb,
                                                      int ldo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * ldo) return;
    const int row = idx / ldo, c = idx % ldo;
    float v = 0.f;
    if (c < D) {
        v = p[row * D + c];
    } else if (c < (2 * L + 1) * D) {
        const int e = c - D, l = e / (2 * D), r = e % (2 * D);
        const float th = __fmul_rn(ldexpf(1.0f, l), p[row * D + (r % D)]);
        v = r < D ? sinf(th) : cosf(th);
    }
    emb[idx] = v;
}

int launch_fourier3d(const float* xyz, int P, int rows_per_pair, const float* R, const float* t, int L, float* emb, int ldo,
                     float* warped_ws, hipStream_t st) {
    if (P <= 0) return DR_OK;
    if (ldo < (2 * L + 1) * 3) return DR_EINVAL;
    ProfScope ps(PK_PE, (double)P * rows_per_pair * (ldo + 6) * 4.0, st);
    hipLaunchKernelGGL(warp_mean_kernel, dim3(P), dim3(256), 0, st, xyz, rows_per_pair, R, t, warped_ws);
    const int total = P * rows_per_pair * ldo;
    hipLaunchKernelGGL(fourier_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)warped_ws, P * rows_per_pair, 3, L, emb, ldo);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

int launch_fourier2d(const float* pix, int rows, int L, float* emb, int ldo, hipStream_t st) {
    if (rows <= 0) return DR_OK;
    if (ldo < (2 * L + 1) * 2) return DR_EINVAL;
    const int total = rows * ldo;
    ProfScope ps(PK_PE, (double)total * 4.0, st);
    hipLaunchKernelGGL(fourier_kernel, dim3((total + 255) / 256), dim3(256), 0, st, pix, rows, 2, L, emb, ldo);
    DR_LAUNCH_CHECK();
    return DR_OK;
}

// ---------------------------------------------------------------------------------------------
// VolumetricPositionEncoding (position_encoding.py:16-23,49-87) of optionally warped points
// (pipeline.py:306  R_forwd * s_pcd + t_forwd).  Tables are stored un-duplicated: [rows, C/2],
// entry (axis a, frequency k) at a*(C/6) + k is the angle of channels 2*(a*C/6 + k), +1.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vol_pe_kernel(const float* __restrict__ xyz, int rows, int rows_per_pair,
                                                     const float* __restrict__ R, const float* __restrict__ tv, int C,
                                                     float ox, float oy, float oz, float voxel,
                                                     const float* __restrict__ freq, float* __restrict__ cosT,
                                                     float* __restrict__ sinT, float* __restrict__ warped, float2* __restrict__ cs_pairs) {
    const int nf = C / 6, half = C / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * half) return;
    const int row = idx / half, e = idx % half;
    const int a = e / nf, k = e % nf;
    float px = xyz[row * 3 + 0], py = xyz[row * 3 + 1], pz = xyz[row * 3 + 2];
    float p;
    if (R) {
        const float* Rp = R + (size_t)(row / rows_per_pair) * 9;
        const float* tp = tv + (size_t)(row / rows_per_pair) * 3;
        // matmul(R, p) + t in the operation order of a 3-term dot product followed by the add
        p = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Rp[a * 3 + 0], px), __fmul_rn(Rp[a * 3 + 1], py)),
                                __fmul_rn(Rp[a * 3 + 2], pz)), tp[a]);
        if (warped && k == 0) warped[row * 3 + a] = p;
    } else {
        p = a == 0 ? px : (a == 1 ? py : pz);
    }
    const float o = a == 0 ? ox : (a == 1 ? oy : oz);
    const float vox = __fdiv_rn(__fsub_rn(p, o), voxel);
    const float ang = __fmul_rn(vox, freq[k]);
    const float cv = cosf(ang), sv = sinf(ang);
    cosT[idx] = cv;
    sinT[idx] = sv;
    if (cs_pairs) cs_pairs[idx] = make_float2(cv, sv);      // the same tables interleaved (cos_k, sin_k): one 16-byte load per rotated float4 in the plane GEMM's epilogue
}

int launch_vol_pe(const float* xyz, int rows, int rows_per_pair, const float* R, const float* t, int C, float ox,
                  float oy, float oz, float voxel, const float* freq, float* cosT, float* sinT, hipStream_t st, float* cs_pairs) {
    if (C % 6) return DR_ENOSUP;
    const int total = rows * (C / 2);
    if (total <= 0) return DR_OK;
    ProfScope ps(PK_PE, (double)total * 8.0, st);
    hipLaunchKernelGGL(vol_pe_kernel, dim3((total + 255) / 256), dim3(256), 0, st, xyz, rows, rows_per_pair, R, t, C, ox,
                       oy, oz, voxel, freq, cosT, sinT, (float*)nullptr, reinterpret_cast<float2*>(cs_pairs));
    DR_LAUNCH_CHECK();
    return DR_OK;
}


// ---------------------------------------------------------------------------------------------
// split_feats (3D/models/pipeline.py:350-379): dst[dst_index[i]][:] = src[src_index[i]][:] for i < n (rows of C floats;
// the padded destination is zero-filled by the caller).  One wave per row.
// ---------------------------------------------------------------------------------------------
// Indices follow torch's indexed assignment: negative ones wrap once ([-rows, -1]); anything else out of range is SKIPPED and
// *status (optional) is set to 1 -- never an out-of-bounds access (PyTorch raises there; the host wrapper turns the flag into one).
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const long long* __restrict__ sidx,
                                                           const long long* __restrict__ didx, float* __restrict__ dst, int n, int C,
                                                           long long n_src, long long n_dst, int* __restrict__ status) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    long long si = sidx[i], di = didx[i];
    if (si < 0) si += n_src;
    if (di < 0) di += n_dst;
    if (si < 0 || si >= n_src || di < 0 || di >= n_dst) {
        if (status && lane == 0) *status = 1;
        return;
    }
    const float* s = src + (size_t)si * C;
    float* d = dst + (size_t)di * C;
    for (int c = lane; c < C; c += 64) d[c] = s[c];
}
}  // namespace dr

extern "C" int dr_scatter_rows_f32(int n, int C, const float* src, int64_t n_src_rows, const int64_t* src_index, const int64_t* dst_index,
                                   float* dst, int64_t n_dst_rows, int32_t* status, void* stream) {
    if (n < 0 || C < 1 || !src || !src_index || !dst_index || !dst || n_src_rows < 0 || n_dst_rows < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    hipLaunchKernelGGL(dr::scatter_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, (const long long*)src_index,
                       (const long long*)dst_index, dst, n, C, (long long)n_src_rows, (long long)n_dst_rows, status);
    DR_LAUNCH_CHECK();
    return DR_OK;
}
