// ViT glue: patch gather (im2col of a stride-14 14x14 conv = pure permutation) and CLS rows.
//   modeling_intern_vit.py:150-152,167-179
#include "misc.hpp"

namespace {

// pixels [T,3,448,448] bf16 -> A [T*1024, 640]; column c*196 + ky*14 + kx (= Conv2d weight.flatten(1) order),
// columns 588..639 zero.  One thread per (patch, segment): 42 segments of 14 contiguous pixels (28 B, 4-B
// aligned on both sides) + 4 zero segments covering the K padding.
__global__ __launch_bounds__(256) void im2col14_kernel(const bf16* __restrict__ px, bf16* __restrict__ out, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int seg = (int)(idx % 46);
    const int64_t m = idx / 46;
    uint32_t* dst = (uint32_t*)(out + m * 640);
    if (seg >= 42) {
        const int w0 = 294 + (seg - 42) * 7;          // 588 bf16 = 294 words; 320 words per row
#pragma unroll
        for (int e = 0; e < 7; e++) if (w0 + e < 320) dst[w0 + e] = 0u;
        return;
    }
    const int t = (int)(m >> 10), pi = (int)(m & 1023), py = pi >> 5, pxx = pi & 31;
    const int c = seg / 14, ky = seg % 14;
    const uint32_t* src = (const uint32_t*)(px + (((int64_t)t * 3 + c) * 448 + (py * 14 + ky)) * 448 + pxx * 14);
    uint32_t v[7];
#pragma unroll
    for (int e = 0; e < 7; e++) v[e] = src[e];
    dst += seg * 7;
#pragma unroll
    for (int e = 0; e < 7; e++) dst[e] = v[e];
}

// x[t*1025][n] = bf16(cls[n] + pos[0][n])
__global__ __launch_bounds__(256) void cls_rows_kernel(const bf16* __restrict__ cls, const bf16* __restrict__ pos,
                                                       bf16* __restrict__ x, int T, int C, int tokens) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * C) return;
    const int t = idx / C, n = idx - t * C;
    x[(int64_t)t * tokens * C + n] = f2bf(bf2f(cls[n]) + bf2f(pos[n]));
}

}  // namespace

int launch_im2col14(const bf16* px, bf16* out, int T, hipStream_t stream) {
    const int64_t total = (int64_t)T * 1024 * 46;
    hipLaunchKernelGGL(im2col14_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, px, out, total);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

int launch_cls_rows(const bf16* cls, const bf16* pos, bf16* x, int T, int C, int tokens, hipStream_t stream) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((T * C + 255) / 256), dim3(256), 0, stream, cls, pos, x, T, C, tokens);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}
