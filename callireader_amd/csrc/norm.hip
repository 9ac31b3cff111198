// LayerNorm / RMSNorm rows for gfx950.  HBM-bound: each thread owns 16 contiguous
// bf16 (two 16-B loads), statistics in fp32 with wave shuffles (+ LDS across
// the 4 waves when a row spans the workgroup), one 16-B store pair per thread.
//
// Reference semantics:
//   nn.LayerNorm(bf16): fp32 mean/var over the row, y = (x-mean)*rstd*gamma+beta rounded once
//     (modeling_intern_vit.py:280-281 eps 1e-6; mlp1[0] modeling_internvl_chat.py:186 eps 1e-5;
//      perceiver_resampler.py:21-22,79,134 eps 1e-5)
//   InternLM2RMSNorm (modeling_internlm2.py:138-143): fp32 x*rsqrt(mean(x^2)+eps) -> bf16 -> * weight
#include "norm.hpp"

namespace {

// TPR threads per row (N = 16 * TPR); 256-thread workgroup holds 256/TPR rows.
template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    if (TPR == 64) return v;
    // TPR == 256: combine the 4 waves
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <int TPR>
__device__ __forceinline__ float row_max(float v, float* red, int tid) {
    v = wave_max(v);
    if (TPR == 64) return v;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// the normalised row as e4m3 + one scale (NormParams::out8): y is rounded to bf16 first, as the bf16 path stores it
template <int TPR>
__device__ __forceinline__ void store_row_e4m3(const NormParams& p, int64_t orow, int t, float* y, float* red, int tid, bool live) {
    float mx = 0.f, ss = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) { y[e] = rbf(y[e]); mx = fmaxf(mx, fabsf(y[e])); ss += y[e] * y[e]; }
    mx = row_max<TPR>(mx, red, tid);
    if (p.next_scale) ss = row_sum<TPR>(ss, red, tid);        // uniform branch
    const float sc = mx > 0.f ? mx / 448.0f : 1.0f;
    if (!live) return;
    store16_e4m3(p.out8 + orow * (16 * TPR) + t * 16, y, 1.0f / sc);
    if (t == 0) {
        p.out8_scale[orow] = sc;
        if (p.next_scale) p.next_scale[orow] = fmaxf((1.13f * sqrtf(ss) * p.next_bound[0] + p.next_bound[1]) / 448.0f, 1e-30f);
    }
}

template <int TPR, int MODE>   // MODE 0: plain rows, 1: pixel-shuffle gather (N = 4096 from [T,1025,1024])
__global__ __launch_bounds__(256) void layernorm_kernel(const NormParams p) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    constexpr int RPB = 256 / TPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + tid / TPR;
    const int t = tid % TPR;
    const bool live = row < p.rows;
    const int64_t r = live ? row : p.rows - 1;
    const bf16* src;
    if (MODE == 1) {
        // out row r = tile*256 + a*16 + b ; features [q*1024 + c], q = t/64:
        //   source token 1 + (2a + (q>>1))*32 + (2b + (q&1))   (modeling_internvl_chat.py:283-297, 311-316)
        const int tile = (int)(r >> 8), ab = (int)(r & 255), a = ab >> 4, b = ab & 15, q = t >> 6;
        const int tok = 1 + (2 * a + (q >> 1)) * 32 + (2 * b + (q & 1));
        src = p.in + ((int64_t)tile * 1025 + tok) * 1024 + (t & 63) * 16;
    } else {
        src = p.in + r * p.ld_in + t * 16;
    }
    float x[16];
    load16(src, x);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) s += x[e];
    const float inv_n = 1.0f / (16 * TPR);
    const float mean = row_sum<TPR>(s, red, tid) * inv_n;
    float v = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) { const float d = x[e] - mean; v += d * d; }
    const float var = row_sum<TPR>(v, red, tid) * inv_n;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    float g[16], bb[16], y[16];
    load16(p.gamma + t * 16, g);
    load16(p.beta + t * 16, bb);
#pragma unroll
    for (int e = 0; e < 16; e++) y[e] = (x[e] - mean) * rstd * g[e] + bb[e];
    if (live) {
        int64_t orow = row;
        if (p.out_group > 0) orow = (row / p.in_group) * p.out_group + p.out_off + row % p.in_group;
        if (!p.out8) store16(p.out + orow * p.ld_out + t * 16, y);
    }
    if (p.out8) {                                             // uniform branch: every thread of the row takes part in the maximum
        int64_t orow = r;
        if (p.out_group > 0) orow = (r / p.in_group) * p.out_group + p.out_off + r % p.in_group;
        store_row_e4m3<TPR>(p, orow, t, y, red, tid, live);
    }
}

__global__ __launch_bounds__(256) void rmsnorm4096_kernel(const NormParams p) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    float x[16], y[16];
    load16(p.in + row * p.ld_in + tid * 16, x);
    rmsnorm_row16(x, p.gamma + tid * 16, p.eps, red, tid, y);
    if (p.out8) store_row_e4m3<256>(p, row, tid, y, red, tid, true);
    else store16(p.out + row * p.ld_out + tid * 16, y);
}

// residual add of a K-sliced GEMM result + the next RMSNorm (decode): modeling_internlm2.py:655-669
//   hidden = residual + attention/ffn output (both bf16) ; hidden -> norm
__global__ __launch_bounds__(256) void add_rmsnorm4096_kernel(bf16* __restrict__ xio, const float* __restrict__ part, int splits,
                                                             int64_t rows, const bf16* __restrict__ gamma, bf16* __restrict__ out,
                                                             float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    float x[16], a[16];
    load16(xio + row * 4096 + tid * 16, x);
#pragma unroll
    for (int e = 0; e < 16; e++) a[e] = 0.f;
    for (int s = 0; s < splits; s++) {
        const f32x4* pp = (const f32x4*)(part + ((int64_t)s * rows + row) * 4096 + tid * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = pp[q];
#pragma unroll
            for (int e = 0; e < 4; e++) a[q * 4 + e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) x[e] = rbf(x[e] + rbf(a[e]));
    store16(xio + row * 4096 + tid * 16, x);
    if (!gamma) return;
    float y[16];
    rmsnorm_row16(x, gamma + tid * 16, eps, red, tid, y);
    store16(out + row * 4096 + tid * 16, y);
}

}  // namespace

int launch_add_rmsnorm(bf16* x, const float* part, int splits, int64_t rows, const bf16* gamma, bf16* out, float eps, hipStream_t stream) {
    if (rows <= 0) return CR_OK;
    if (!x || !part || splits <= 0 || (gamma && !out)) return CR_ERR_ARG;
    hipLaunchKernelGGL(add_rmsnorm4096_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, part, splits, rows, gamma, out, eps);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

int launch_layernorm(const NormParams& p, int n, int mode, hipStream_t stream) {
    if (p.rows <= 0) return CR_OK;
    if (mode == 1) {
        if (n != 4096) return CR_ERR_ARG;
        hipLaunchKernelGGL((layernorm_kernel<256, 1>), dim3((unsigned)p.rows), dim3(256), 0, stream, p);
    } else if (n == 1024) {
        hipLaunchKernelGGL((layernorm_kernel<64, 0>), dim3((unsigned)((p.rows + 3) / 4)), dim3(256), 0, stream, p);
    } else if (n == 4096) {
        hipLaunchKernelGGL((layernorm_kernel<256, 0>), dim3((unsigned)p.rows), dim3(256), 0, stream, p);
    } else {
        return CR_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

int launch_rmsnorm(const NormParams& p, int n, hipStream_t stream) {
    if (p.rows <= 0) return CR_OK;
    if (n != 4096) return CR_ERR_ARG;
    hipLaunchKernelGGL(rmsnorm4096_kernel, dim3((unsigned)p.rows), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}
