// Epilogues shared by the tiled GEMM kernels: a wave's 64x64 fp32 sub-tile sits in its private LDS stage
// (STAGE_LD floats per row); each lane owns 8 contiguous columns of a row -> 16-B loads/stores, the
// reference's bf16 rounding sequence.
#pragma once
#include "common.hpp"

constexpr int STAGE_LD = 68;      // fp32 words per staged row (64 + 4 pad)

template <int EPI>
__device__ __forceinline__ void epilogue_row8(const GemmParams& p, int gm, int gn, const float* v) {
    // v[0..7]: fp32 accumulators of row gm, columns gn..gn+7 (gn % 8 == 0, gn + 8 <= N)
    float x[8];
    // EPI_STORE / EPI_PATCH-free paths whose only rounding is the final cast skip the explicit round-trip
    constexpr bool ROUND_NOW = (EPI != EPI_STORE);
    if (p.bias) {
        bf16x8 b = *(const bf16x8*)(p.bias + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = ROUND_NOW ? rbf(v[e] + bf2f(b[e])) : v[e] + bf2f(b[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = ROUND_NOW ? rbf(v[e]) : v[e];
    }
    if (EPI == EPI_F32) {
        float* c = (float*)p.C + (int64_t)gm * p.ldc + gn;
        if ((p.ldc & 3) == 0) {
            *(f32x4*)c = f32x4{x[0], x[1], x[2], x[3]};
            *(f32x4*)(c + 4) = f32x4{x[4], x[5], x[6], x[7]};
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) c[e] = x[e];
        }
        return;
    }
    int64_t orow = gm;
    if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = gelu_erf(x[e]);
    } else if (EPI == EPI_LS_RES) {
        bf16x8 s = *(const bf16x8*)(p.scale + gn);
        bf16x8 r = *(const bf16x8*)(p.res + (int64_t)gm * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = bf2f(r[e]) + rbf(x[e] * bf2f(s[e]));
    } else if (EPI == EPI_RES) {
        bf16x8 r = *(const bf16x8*)(p.res + (int64_t)gm * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = bf2f(r[e]) + x[e];
    } else if (EPI == EPI_PATCH) {
        int t = gm / p.group, pi = gm - t * p.group;
        orow = (int64_t)t * (p.group + 1) + 1 + pi;
        bf16x8 r = *(const bf16x8*)(p.res + (int64_t)(1 + pi) * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = x[e] + bf2f(r[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(x[e]);
    *(bf16x8*)((bf16*)p.C + orow * p.ldc + gn) = o;
}

template <int EPI>
__device__ __forceinline__ void epilogue_scalar(const GemmParams& p, int gm, int gn, float v) {
    float x = rbf(v + (p.bias ? bf2f(p.bias[gn]) : 0.0f));
    if (EPI == EPI_F32) { ((float*)p.C)[(int64_t)gm * p.ldc + gn] = x; return; }
    int64_t orow = gm;
    if (EPI == EPI_GELU) x = gelu_erf(x);
    else if (EPI == EPI_LS_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn]) + rbf(x * bf2f(p.scale[gn]));
    else if (EPI == EPI_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn]) + x;
    else if (EPI == EPI_PATCH) {
        int t = gm / p.group, pi = gm - t * p.group;
        orow = (int64_t)t * (p.group + 1) + 1 + pi;
        x = x + bf2f(p.res[(int64_t)(1 + pi) * p.ldr + gn]);
    }
    ((bf16*)p.C)[orow * p.ldc + gn] = f2bf(x);
}


// rows [row0, row0+64) x cols [col0, col0+64) of the output from the wave-private stage `st`
template <int EPI>
__device__ __forceinline__ void epilogue_subtile(const GemmParams& p, const float* st, int row0, int col0, int lane) {
    if (EPI == EPI_ARGMAX) {
        // lane = row of the 64 x 64 sub-tile: first maximum of bf16(acc + bias) over its 64 columns -> one 8-byte partial
        const int gm = row0 + lane;
        float bv = -INFINITY; int bc = 0x7fffffff;
        for (int c = 0; c < 64; c++) {
            const int n = col0 + c;
            float v = rbf(st[lane * STAGE_LD + c] + (p.bias && n < p.N ? bf2f(p.bias[n]) : 0.f));
            v = n < p.N ? v : -INFINITY;
            if (v > bv) { bv = v; bc = n; }
        }
        if (gm < p.M && col0 < p.N) ((unsigned long long*)p.C)[(int64_t)gm * p.ldc + (col0 >> 6)] = argmax_pack(bv, bc);
        return;
    }
    if (EPI == EPI_SWIGLU) {
        // staged columns: [8 gate | 8 up] x 4 per 64-wide row -> 32 outputs per row
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 16 + (lane >> 2);
            const int oc = lane & 3;
            const int gm = row0 + row;
            const int gno = col0 / 2 + oc * 8;
            if (gm < p.M && gno + 8 <= p.N / 2) {
                const float* sp = st + row * STAGE_LD + oc * 16;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float g = rbf(sp[e]), u = rbf(sp[8 + e]);
                    o[e] = f2bf(rbf(silu(g)) * u);
                }
                *(bf16x8*)((bf16*)p.C + (int64_t)gm * p.ldc + gno) = o;
            }
        }
        return;
    }
    const bool vec_ok = ((p.ldc & 7) == 0) || (EPI == EPI_F32);
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int row = it * 8 + (lane >> 3);
        const int c8 = (lane & 7) * 8;
        const int gm = row0 + row;
        const int gn = col0 + c8;
        if (gm >= p.M || gn >= p.N) continue;
        const float* sp = st + row * STAGE_LD + c8;
        if (gn + 8 <= p.N && vec_ok) {
            f32x4 v0 = *(const f32x4*)sp, v1 = *(const f32x4*)(sp + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            epilogue_row8<EPI>(p, gm, gn, v);
        } else {
            for (int e = 0; e < 8 && gn + e < p.N; e++) epilogue_scalar<EPI>(p, gm, gn + e, sp[e]);
        }
    }
}

