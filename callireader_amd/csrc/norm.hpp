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
