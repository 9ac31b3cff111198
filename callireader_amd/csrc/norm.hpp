#pragma once
#include "common.hpp"

struct NormParams {
    const bf16* in; int64_t ld_in;
    bf16* out; int64_t ld_out;
    const bf16* gamma;
    const bf16* beta;          // nullptr for RMSNorm
    int64_t rows;
    float eps;
    // optional output row regrouping: out_row = (row / in_group) * out_group + out_off + row % in_group
    int in_group, out_group, out_off;
    // fp8 matrix-core path: when out8 is set the row leaves as e4m3 bytes out8[row][n] (leading dimension n) with one fp32
    // scale per row, out8_scale[row] = max|y| / 448 over the bf16-rounded normalised row y (1 for an all-zero row), q = RNE(y / scale);
    // `out` is then not written (the normalised row only feeds the next GEMM)
    unsigned char* out8; float* out8_scale;
    // with out8: next_scale[row] = (1.13 * ||y||_2 * next_bound[0] + next_bound[1]) / 448, a bound on |y~ . w~_n + b_n| for every output n
    // of the linear that follows (Cauchy-Schwarz; next_bound = {largest weight-row norm, largest |bias|} on the device, 1.13 = the
    // slack of both operands' e4m3 rounding, 1.0625^2) -- the e4m3 scale that linear's epilogue may use for ITS output row without
    // knowing the row's maximum (EPI_GELU_Q8)
    float* next_scale; const float* next_bound;
};

// mode 0: rows of n (1024 | 4096) bf16; mode 1: pixel-shuffle gather feeding mlp1's LayerNorm (n = 4096)
int launch_layernorm(const NormParams& p, int n, int mode, hipStream_t stream);
int launch_rmsnorm(const NormParams& p, int n, hipStream_t stream);
// decode: x[row] = bf16(x[row] + bf16(sum_s part[s][row])) for rows of 4096 (part = fp32 [S][rows][4096], the K-slices of
// an EPI_PARTIAL GEMM, summed in slice order), then, when gamma is given, out[row] = RMSNorm(x[row]) * gamma
int launch_add_rmsnorm(bf16* x, const float* part, int splits, int64_t rows, const bf16* gamma, bf16* out, float eps, hipStream_t stream);

// ---- shared by norm.hip and gemm_decode.hip (the same code, so the same bits) -------------------------------------------------------------
__device__ __forceinline__ void load16(const bf16* p, float* x) {
    bf16x8 a = *(const bf16x8*)p, b = *(const bf16x8*)(p + 8);
#pragma unroll
    for (int e = 0; e < 8; e++) { x[e] = bf2f(a[e]); x[8 + e] = bf2f(b[e]); }
}
__device__ __forceinline__ void store16(bf16* p, const float* y) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; e++) { a[e] = f2bf(y[e]); b[e] = f2bf(y[8 + e]); }
    *(bf16x8*)p = a; *(bf16x8*)(p + 8) = b;
}
// InternLM2RMSNorm (modeling_internlm2.py:138-143) on a row of 4096 in three pieces, shared by every kernel that normalises a row (the same
// code, so the same bits).  A "thread slot" t of the row's 256 holds x[16] = elements 16t .. 16t+15.
//   rms_sumsq16: the slot's sum of squares, squares and sums as separate operations as in the reference's x.pow(2).mean() (no FMA contraction);
//   the row total is ((W0 + W1) + W2) + W3 with Wq = wave_sum over slots 64q .. 64q+63 (butterfly over the lane index);
//   rms_apply16: y = weight * bf16(x * rsqrt(total / 4096 + eps)) as fp32 values (the caller rounds on store).
__device__ __forceinline__ float rms_sumsq16(const float (&x)[16]) {
#pragma clang fp contract(off)
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) ss += x[e] * x[e];
    return ss;
}
__device__ __forceinline__ void rms_apply16(const float (&x)[16], float total, float eps, const bf16* gamma16, float (&y)[16]) {
#pragma clang fp contract(off)
    const float var = total * (1.0f / 4096.0f);
    const float rs = rsqrtf(var + eps);
    float g[16];
    load16(gamma16, g);
#pragma unroll
    for (int e = 0; e < 16; e++) y[e] = g[e] * rbf(x[e] * rs);
}
// 256 threads per row (norm.hip's layout): thread t = slot t.  red4: four floats of LDS of this row's thread group.  EVERY thread of the
// workgroup must make the call (two __syncthreads()).
__device__ __forceinline__ void rmsnorm_row16(const float (&x)[16], const bf16* gamma16, float eps, float* red4, int t, float (&y)[16]) {
    const float v = wave_sum(rms_sumsq16(x));
    __syncthreads();
    if ((t & 63) == 0) red4[t >> 6] = v;
    __syncthreads();
    const float tot = red4[0] + red4[1] + red4[2] + red4[3];
    rms_apply16(x, tot, eps, gamma16, y);
}
