// Epilogue of one wave's 128 x 64 accumulator sub-tile (8 x 4 tiles of v_mfma_f32_16x16x32_bf16 with the weights as the
// A operand), shared by the 8-wave 256x256 kernel (gemm256.hip) and the 4-wave 256x128 kernel (gemm2w.hip).
#pragma once
#include "gemm_epilogue.hpp"

namespace {

// ---- epilogue of one wave's 128 x 64 sub-tile ---------------------------------------------------------------------
// The MFMAs run with the operands swapped (weights as the A operand), so a lane's accumulator acc[mf][j][0..3] is
// row m = mf*16 + (lane & 15), columns n = j*16 + (lane >> 4)*4 + 0..3: four CONSECUTIVE columns.  Everything that is
// per element and per column is applied right there (bias, the bf16 rounding of the linear output, GELU, LayerScale),
// the four values are packed to bf16 and leave as ONE 8-byte ds_write into a 16 x 64 bf16 slice (2 KiB, two slices
// per wave so that staging slice mf + 1 does not wait for slice mf's read-back).  The read-back hands every lane 8
// consecutive columns of a row (16 B): residual / position rows are added there and the result goes out as 16-byte
// stores.  fp32 staging through ds_write_b32 (64 B/clk/CU for all 8 waves) used to bound the epilogue at ~4 k cycles.
//   slice image: row r at r*128 B; the 8-byte slot s of a row sits at (s ^ r): the 16 lanes of a ds_write_b64 group
//   (rows 0..15, same s) hit 16 different bank pairs, and the 16-byte read-back chunk c = (slot pair) is found at
//   c ^ (r >> 1) with its halves exchanged when r is odd.
template <int EPI>
__device__ __forceinline__ void stage_slice(const f32x4 (&a)[4], const float (&bias_f)[4][4], const float (&scale_f)[4][4],
                                            bool has_bias, char* buf, int lane) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float x[4] = {a[j][0], a[j][1], a[j][2], a[j][3]};
        if (has_bias) {
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] += bias_f[j][e];
        }
        if (EPI == EPI_GELU || EPI == EPI_LS_RES) {          // bf16(acc + bias) is a value of its own before the next op
            round_pair_bf16(x[0], x[1], x[0], x[1]);
            round_pair_bf16(x[2], x[3], x[2], x[3]);
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] = EPI == EPI_GELU ? gelu_erf(x[e]) : x[e] * scale_f[j][e];
        }
        const bf16x4 o = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3])};
        *(bf16x4*)(buf + r * 128 + (((j * 4 + g) ^ r) << 3)) = o;
    }
}

// 16-byte chunk c (columns 8c .. 8c+7) of staged row r
__device__ __forceinline__ bf16x8 read_chunk(const char* buf, int r, int c) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 v = *(const u32x4*)(buf + r * 128 + ((c ^ (r >> 1)) << 4));
    const bool odd = r & 1;
    const u32x4 w = {odd ? v[2] : v[0], odd ? v[3] : v[1], odd ? v[0] : v[2], odd ? v[1] : v[3]};
    return __builtin_bit_cast(bf16x8, w);
}

// NSLICE = 2-KiB staging slices per wave: 2 lets staging slice mf + 1 start while slice mf is read back, 1 halves the LDS
template <int EPI, int NSLICE = 2>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, const f32x4 (&acc)[8][4], char* stg, int row_base, int col0,
                                              int lane) {
    constexpr bool ADD_ROWS = (EPI == EPI_LS_RES || EPI == EPI_RES);
    const bool has_bias = p.bias != nullptr;
    const bool full_n = col0 + 64 <= p.N;
    // ---- per-column operands in the accumulator layout
    float bias_f[4][4], scale_f[4][4];
    {
        const bool vec = full_n && ((((uintptr_t)p.bias) | ((uintptr_t)p.scale)) & 7) == 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = col0 + j * 16 + (lane >> 4) * 4;
            if (has_bias) {
                if (vec) {
                    const bf16x4 b = *(const bf16x4*)(p.bias + n);
#pragma unroll
                    for (int e = 0; e < 4; e++) bias_f[j][e] = bf2f(b[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) bias_f[j][e] = n + e < p.N ? bf2f(p.bias[n + e]) : 0.f;
                }
            }
            if (EPI == EPI_LS_RES) {
                if (vec) {
                    const bf16x4 b = *(const bf16x4*)(p.scale + n);
#pragma unroll
                    for (int e = 0; e < 4; e++) scale_f[j][e] = bf2f(b[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) scale_f[j][e] = n + e < p.N ? bf2f(p.scale[n + e]) : 0.f;
                }
            }
        }
    }

    if (EPI == EPI_ARGMAX) {
        // cosine VQ (models/similarity.py:19-21): per row the maximum of bf16(acc) over this wave's 64 columns and its
        // first column; a lane holds 16 of a row's 64 values (columns j*16 + g*4 + e), the four lane groups g meet through
        // two shuffles.  One 8-byte partial per (row, 64-column block) leaves the kernel; the similarity never does.
        const int r = lane & 15, g = lane >> 4;
        const int64_t blk = col0 >> 6;
#pragma unroll
        for (int mf = 0; mf < 8; mf++) {
            float bv = -INFINITY; int bc = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int n = col0 + j * 16 + g * 4 + e;
                    float v = rbf(acc[mf][j][e] + (has_bias ? bias_f[j][e] : 0.f));
                    v = n < p.N ? v : -INFINITY;
                    if (v > bv) { bv = v; bc = n; }              // ascending columns inside a lane: strict > keeps the first
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) argmax_merge(bv, bc, __shfl_xor(bv, o, 64), __shfl_xor(bc, o, 64));
            const int gm = row_base + mf * 16 + r;
            if (g == 0 && gm < p.M && col0 < p.N) ((unsigned long long*)p.C)[(int64_t)gm * p.ldc + blk] = argmax_pack(bv, bc);   // blocks past N do not exist
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
        // staged columns are [8 gate | 8 up] per 16: a lane takes one such pair -> 8 outputs
        const int r = lane >> 2, oc = lane & 3;
        const int gno = col0 / 2 + oc * 8;
#pragma unroll
        for (int mf = 0; mf < 8; mf++) {
            char* buf = stg + (mf & (NSLICE - 1)) * 2048;
            stage_slice<EPI>(acc[mf], bias_f, scale_f, has_bias, buf, lane);
            __builtin_amdgcn_wave_barrier();
            const int gm = row_base + mf * 16 + r;
            const bf16x8 gt = read_chunk(buf, r, 2 * oc), up = read_chunk(buf, r, 2 * oc + 1);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = f2bf(rbf(silu(bf2f(gt[e]))) * bf2f(up[e]));
            if (gm < p.M && gno + 8 <= p.N / 2) __builtin_nontemporal_store(o, (bf16x8*)((bf16*)p.C + (int64_t)gm * p.ldc + gno));
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }

    const int rin = lane >> 3, c = lane & 7, gn = col0 + c * 8;
    const bool vec_ok = gn + 8 <= p.N && (EPI == EPI_F32 ? (p.ldc & 3) == 0 : (p.ldc & 7) == 0) && (p.ldr & 7) == 0;
    // rows to add after the read-back (residual stream / position embedding), requested one slice ahead
    bf16x8 radd[2], rnext[2];
    auto row_of = [&](int mf, int it) { return row_base + mf * 16 + it * 8 + rin; };
    auto load_rows = [&](int mf, bf16x8 (&dst)[2]) {
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int gm = min(row_of(mf, it), p.M - 1);
            if (EPI == EPI_PATCH) dst[it] = *(const bf16x8*)(p.res + (int64_t)(1 + gm % p.group) * p.ldr + gn);
            else dst[it] = *(const bf16x8*)(p.res + (int64_t)gm * p.ldr + gn);
        }
    };
    constexpr bool PRE = ADD_ROWS || EPI == EPI_PATCH;
    if (PRE && vec_ok) load_rows(0, radd);
#pragma unroll
    for (int mf = 0; mf < 8; mf++) {
        char* buf = stg + (mf & (NSLICE - 1)) * 2048;
        stage_slice<EPI>(acc[mf], bias_f, scale_f, has_bias, buf, lane);
        __builtin_amdgcn_wave_barrier();
        if (PRE && vec_ok && mf + 1 < 8) load_rows(mf + 1, rnext);
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int r = it * 8 + rin;
            const int gm = row_of(mf, it);
            const bf16x8 v = read_chunk(buf, r, c);
            if (gm >= p.M || gn >= p.N) continue;
            int64_t orow = gm;
            if (EPI == EPI_PATCH) { const int t = gm / p.group; orow = (int64_t)t * (p.group + 1) + 1 + (gm - t * p.group); }
            if (vec_ok) {
                if (EPI == EPI_F32) {
                    float* cp = (float*)p.C + orow * p.ldc + gn;
                    *(f32x4*)cp = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
                    *(f32x4*)(cp + 4) = f32x4{bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7])};
                } else if (PRE) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(radd[it][e]) + bf2f(v[e]));
                    __builtin_nontemporal_store(o, (bf16x8*)((bf16*)p.C + orow * p.ldc + gn));
                } else {
                    __builtin_nontemporal_store(v, (bf16x8*)((bf16*)p.C + orow * p.ldc + gn));
                }
            } else {                                          // ragged N or unaligned rows: element by element
                for (int e = 0; e < 8 && gn + e < p.N; e++) {
                    float x = bf2f(v[e]);
                    if (ADD_ROWS) x = bf2f(p.res[(int64_t)gm * p.ldr + gn + e]) + x;
                    if (EPI == EPI_PATCH) x = x + bf2f(p.res[(int64_t)(1 + gm % p.group) * p.ldr + gn + e]);
                    if (EPI == EPI_F32) ((float*)p.C)[orow * p.ldc + gn + e] = x;
                    else ((bf16*)p.C)[orow * p.ldc + gn + e] = f2bf(x);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (PRE) { radd[0] = rnext[0]; radd[1] = rnext[1]; }
    }
}


}  // namespace
