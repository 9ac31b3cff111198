// Weight-streaming GEMM for decode:  C[M,N] = epi(X[M,K] . W[N,K]^T),  M <= 64 rows.
//
// HBM-bound: every W byte is read exactly once, X (M x K, <= 1.8 MB) is served from L2.  No LDS staging
// ("GEMV / M <= 16 decode weights: load straight to VGPRs, deep unroll, late vmcnt").
//   * one workgroup owns 16 rows of W (one v_mfma_f32_16x16x32_bf16 A-tile) over the whole K;
//     its WAVES waves split K into contiguous slices, so N/16 * WAVES waves keep enough loads in flight
//     to cover HBM latency (WAVES = 8 for N <= 8192, 4 above);
//   * RT = 16-row weight tiles per wave (4 where N >= 16384 and M > 16): the X fragments are loaded once per RT
//     tiles; X comes from L2 but through the same per-CU load path as W, and at M = 32 it is 2 x the W bytes at RT = 1;
//   * per k-step a lane loads 16 B of W (row n = lane&15, k = 8*(lane>>4) .. +7) and 16 B of each X tile,
//     UNROLL k-steps of loads are issued before the first MFMA; with p.wsw the weights come from their DECODE-LAYOUT copy (gemm_decode.hip:
//     decode_swizzle_kernel -- the 64 lanes' fragments of a tile and k-step side by side), so a load instruction is one contiguous KiB instead of
//     16 rows x 64 bytes: wqkv / wo / w2 at 64 rows 19.6 / 14.6 / 37.7 -> 17.2 / 13.6 / 29.7 us, the same bits;
//   * accumulator = C^T tile [16 n][16 m] per X tile, so the lane holds 4 consecutive n of one m;
//   * partial tiles of the waves are summed through LDS in a fixed order (bitwise reproducible, and a row's
//     result does not depend on how many other rows are in the batch);
//   * epilogues: store / +residual / SwiGLU (rows [8 gate | 8 up] of one tile -> 8 outputs) / fp32 logits.
//   * W8: the weights are e4m3 bytes with one fp32 scale per row (llm.hip: cr_enable_fp8_decode) -- half the HBM bytes of the
//     kernel's bound.  A lane loads 16 B = 16 consecutive k of its row per 64-deep step and turns them into two bf16
//     fragments with v_cvt_scalef32_pk_bf16_fp8 (exact: e4m3 is a subset of bf16); the X fragments take the same k
//     (16*(lane>>4) + 0..7 and + 8..15), the MFMA, the accumulation and the epilogue stay as they are, and the row scale
//     multiplies the fp32 sum once, before the epilogue's rounding.
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"
#include "diag.hpp"

// gemm_stream.hip: X shared through LDS, waves split N (w1|w3, LM head at more than 16 rows); the same fp32 sums as launch_w's four-wave form
bool gemm_stream_supported(int epi, const GemmParams& p, int splits);
int launch_gemm_stream(int epi, const GemmParams& p, hipStream_t stream, int splits);

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// 8 e4m3 bytes (two dwords) -> one bf16x8 MFMA fragment
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(unsigned lo, unsigned hi) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    return bf16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

template <int EPI, int WAVES, int MT, int RT = 1, bool W8 = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(const GemmParams p) {
    // RT = 16-row weight tiles per wave: X fragments are loaded once per RT tiles (X re-reads through L2 are the
    // bottleneck once M > 16), used where N is large enough to still fill the chip
    __shared__ float red[WAVES][RT * MT][16][17];
    constexpr int UNROLL = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16 * RT;
    const int kq = (lane >> 4) * 8;
    // gridDim.y K-slices (EPI_PARTIAL; 1 otherwise), each split over the waves
    const int ksteps = p.K / (32 * WAVES * (int)gridDim.y);            // k-steps of 32 per wave
    const int kbase = ((int)blockIdx.y * WAVES + wave) * ksteps * 32;
    const bf16* wp[RT];
    // decode layout (p.wsw): tile (n0 / 16 + r), k-step kbase / 32 onwards, one KiB per k-step, this lane's 16 bytes at lane * 16
    const int wstep = (!W8 && p.wsw) ? 512 : 32;                  // elements between a lane's loads of consecutive k-steps
#pragma unroll
    for (int r = 0; r < RT; r++)
        wp[r] = (!W8 && p.wsw) ? p.W + ((int64_t)min(n0 / 16 + r, (p.N + 15) / 16 - 1) * (p.K / 32) + kbase / 32) * 512 + lane * 8
                               : p.W + (int64_t)min(n0 + r * 16 + (lane & 15), p.N - 1) * p.ldw + kbase + kq;
    const bf16* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) xp[t] = p.A + (int64_t)min(t * 16 + (lane & 15), p.M - 1) * p.lda + kbase + kq;
#ifdef CR_KO_XFRAG      // knock-out (wrong results, cost structure only): the activations read as if stored in fragment layout, a contiguous KiB per load instruction
    constexpr int XSTEP = 512;
#pragma unroll
    for (int t = 0; t < MT; t++) xp[t] = p.A + (int64_t)min(t * 16, max(p.M - 16, 0)) * p.lda + (kbase / 32) * 512 + lane * 8;
#else
    constexpr int XSTEP = 32;
#endif

    constexpr int UR = UNROLL / RT;
    f32x4 acc[RT][MT];
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (W8) {
        // 64-deep steps: per step a lane holds k = 16*(lane>>4) + 0..15 of its weight row (one 16-B load) and of each X row
        // (two 16-B loads); fragment h (h = 0, 1) pairs the weights' bytes 8h..8h+7 with the X chunk at + 8h
        const int steps = ksteps / 2;
        const int kq16 = (lane >> 4) * 16;
        const unsigned char* w8p[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) w8p[r] = (const unsigned char*)p.W + (int64_t)min(n0 + r * 16 + (lane & 15), p.N - 1) * p.ldw + kbase + kq16;
        // decode layout of the e4m3 copy (p.wsw, gemm_decode.hip: decode_swizzle8_kernel): a load instruction is one contiguous KiB instead of 16 rows x 64 bytes
        // (64 rows: a step 9.76 -> 8.83 ms when this was a knock-out in round 4, 16 rows 4.53 -> 3.91)
        const int W8STEP = p.wsw ? 1024 : 64;
        if (p.wsw) {
#pragma unroll
            for (int r = 0; r < RT; r++) w8p[r] = (const unsigned char*)p.W + ((int64_t)min(n0 / 16 + r, (p.N + 15) / 16 - 1) * (p.K / 64) + kbase / 64) * 1024 + lane * 16;
        }
        const bf16* x8p[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) x8p[t] = p.A + (int64_t)min(t * 16 + (lane & 15), p.M - 1) * p.lda + kbase + kq16;
        constexpr int U8 = UR > 1 ? UR / 2 : 1;               // 64-deep steps in flight per batch
        int s8 = 0;
        for (; s8 + U8 <= steps; s8 += U8) {
            u32x4_t w[RT][U8];
            bf16x8 x[MT][U8][2];
#pragma unroll
            for (int r = 0; r < RT; r++)
#pragma unroll
                for (int u = 0; u < U8; u++) w[r][u] = __builtin_nontemporal_load((const u32x4_t*)(w8p[r] + (s8 + u) * W8STEP));
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int u = 0; u < U8; u++) {
                    x[t][u][0] = *(const bf16x8*)(x8p[t] + (s8 + u) * 64);
                    x[t][u][1] = *(const bf16x8*)(x8p[t] + (s8 + u) * 64 + 8);
                }
#pragma unroll
            for (int u = 0; u < U8; u++)
#pragma unroll
                for (int r = 0; r < RT; r++) {
                    const bf16x8 w0 = fp8x8_to_bf16(w[r][u][0], w[r][u][1]), w1 = fp8x8_to_bf16(w[r][u][2], w[r][u][3]);
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x[t][u][0], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x[t][u][1], acc[r][t], 0, 0, 0);
                    }
                }
        }
        for (; s8 < steps; s8++) {
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const u32x4_t w = __builtin_nontemporal_load((const u32x4_t*)(w8p[r] + s8 * W8STEP));
                const bf16x8 w0 = fp8x8_to_bf16(w[0], w[1]), w1 = fp8x8_to_bf16(w[2], w[3]);
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, *(const bf16x8*)(x8p[t] + s8 * 64), acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, *(const bf16x8*)(x8p[t] + s8 * 64 + 8), acc[r][t], 0, 0, 0);
                }
            }
        }
    }
    int ks = W8 ? ksteps : 0;
    if (WAVES == 1 && RT == 1 && !W8) {
        // The tail kernel (launch_gemm_tail): ONE wave walks the whole K in ascending order (the tiled kernels' order: these rows must
        // get their bits), so its time is a chain of load round trips -- 128 k-steps at K = 4096.  Two register batches of UR k-steps:
        // the loads of batch b + 1 are in flight while batch b is multiplied (config 2's fc2 tail: 43 us per launch before).
        bf16x8 wa[UR], wb[UR], xa[MT][UR], xb[MT][UR];
        auto load = [&](bf16x8* w, bf16x8 (*x)[UR], int k0) {
#pragma unroll
            for (int u = 0; u < UR; u++) w[u] = __builtin_nontemporal_load((const bf16x8*)(wp[0] + (k0 + u) * 32));
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int u = 0; u < UR; u++) x[t][u] = *(const bf16x8*)(xp[t] + (k0 + u) * 32);
        };
        auto mma = [&](const bf16x8* w, const bf16x8 (*x)[UR]) {
#pragma unroll
            for (int u = 0; u < UR; u++)
#pragma unroll
                for (int t = 0; t < MT; t++) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u], x[t][u], acc[0][t], 0, 0, 0);
        };
        const int nb = ksteps / UR;
        if (nb > 0) {
            load(wa, xa, 0);
            int b = 0;
            for (; b + 2 <= nb; b += 2) {                    // (wa, xa) hold batch b
                load(wb, xb, (b + 1) * UR);
                mma(wa, xa);
                if (b + 2 < nb) load(wa, xa, (b + 2) * UR);
                mma(wb, xb);
            }
            if (b < nb) mma(wa, xa);
            ks = nb * UR;
        }
    }
    for (; ks + UR <= ksteps; ks += UR) {
        bf16x8 w[RT][UR], x[MT][UR];
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int u = 0; u < UR; u++) w[r][u] = __builtin_nontemporal_load((const bf16x8*)(wp[r] + (ks + u) * wstep));
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int u = 0; u < UR; u++) x[t][u] = *(const bf16x8*)(xp[t] + (ks + u) * XSTEP);
#pragma unroll
        for (int u = 0; u < UR; u++)
#pragma unroll
            for (int r = 0; r < RT; r++)
#pragma unroll
                for (int t = 0; t < MT; t++) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[r][u], x[t][u], acc[r][t], 0, 0, 0);
    }
    for (; ks < ksteps; ks++) {
        bf16x8 x[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) x[t] = *(const bf16x8*)(xp[t] + ks * 32);
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const bf16x8 w = __builtin_nontemporal_load((const bf16x8*)(wp[r] + ks * wstep));
#pragma unroll
            for (int t = 0; t < MT; t++) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x[t], acc[r][t], 0, 0, 0);
        }
    }
    // C^T tile: row (n) = (lane>>4)*4 + e, col (m) = lane&15; staged tile index = r * MT + t
#pragma unroll
    for (int r = 0; r < RT; r++)
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int e = 0; e < 4; e++) red[wave][r * MT + t][(lane >> 4) * 4 + e][lane & 15] = acc[r][t][e];
    __syncthreads();

    // thread -> (m tile, n, m); sum the waves' partials in wave order
    for (int idx = tid; idx < RT * MT * 256; idx += WAVES * 64) {
        const int t = idx >> 8, n = idx & 15, m = (idx >> 4) & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; w++) s += red[w][t][n][m];
        if (W8) s *= p.wscale[min(n0 + (t / MT) * 16 + n, p.N - 1)];      // fp8 weights: the row's scale, once, on the fp32 sum
        red[0][t][n][m] = s;       // own element only: no hazard with other threads' reads
    }
    __syncthreads();
    if (EPI == EPI_PARTIAL) {
        float* out = (float*)p.C + (int64_t)blockIdx.y * p.M * p.ldc;
        for (int idx = tid; idx < RT * MT * 256; idx += WAVES * 64) {
            const int t = idx >> 8, n = idx & 15, m = (idx >> 4) & 15;
            const int gm = (t % MT) * 16 + m, gn = p.wsw == 2 ? rope_tile_row(n0 / 16 + t / MT, n) : n0 + (t / MT) * 16 + n;
            if (gm < p.M && gn < p.N) out[(int64_t)gm * p.ldc + gn] = red[0][t][n][m];
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
        for (int idx = tid; idx < RT * MT * 128; idx += WAVES * 64) {
            const int q = idx >> 7, j = idx & 7, m = (idx >> 3) & 15;
            const int t = q;
            const int gm = (q % MT) * 16 + m, gno = (n0 + (q / MT) * 16) / 2 + j;
            if (gm < p.M && gno < p.N / 2) {
                const float g = rbf(red[0][t][j][m]), u = rbf(red[0][t][8 + j][m]);
                ((bf16*)p.C)[(int64_t)gm * p.ldc + gno] = f2bf(rbf(silu(g)) * u);
            }
        }
        return;
    }
    for (int idx = tid; idx < RT * MT * 256; idx += WAVES * 64) {
        const int t = idx >> 8, n = idx & 15, m = (idx >> 4) & 15;
        const int gm = (t % MT) * 16 + m, gn = n0 + (t / MT) * 16 + n;
        if (gm >= p.M || gn >= p.N) continue;
        float x = rbf(red[0][t][n][m] + (p.bias ? bf2f(p.bias[gn]) : 0.f));
        if (EPI == EPI_F32) { ((float*)p.C)[(int64_t)gm * p.ldc + gn] = x; continue; }
        if (EPI == EPI_GELU) x = gelu_erf(x);
        if (EPI == EPI_LS_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn]) + rbf(x * bf2f(p.scale[gn]));
        if (EPI == EPI_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn]) + x;
        ((bf16*)p.C)[(int64_t)gm * p.ldc + gn] = f2bf(x);
    }
}

template <int EPI, int WAVES, int RT, bool W8>
int launch_mt8(const GemmParams& p, hipStream_t stream, int splits) {
    const int mt = (p.M + 15) / 16;
    const dim3 grid((p.N + 16 * RT - 1) / (16 * RT), splits), block(WAVES * 64);
    switch (mt) {
        case 1: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, WAVES, 1, RT, W8>), grid, block, 0, stream, p); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, WAVES, 2, RT, W8>), grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, WAVES, 3, RT, W8>), grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, WAVES, 4, RT, W8>), grid, block, 0, stream, p); break;
        default: return CR_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

template <int EPI, int WAVES, int RT>
int launch_mt(const GemmParams& p, hipStream_t stream, int splits = 1) {
    if (p.w8) {
        if (p.K % (64 * WAVES * splits) != 0 || !p.wscale) return CR_ERR_ARG;      // 64-deep steps per wave
        return launch_mt8<EPI, WAVES, RT, true>(p, stream, splits);
    }
    return launch_mt8<EPI, WAVES, RT, false>(p, stream, splits);
}

template <int EPI>
int launch_w(const GemmParams& p, hipStream_t stream) {
    if (p.N <= 8192 && p.K % 256 == 0) return launch_mt<EPI, 8, 1>(p, stream);
    // four row tiles per wave where N still fills the chip: a quarter of the X re-reads (w1|w3 at M = 32: 81 us with
    // two tiles, 63 us with four, 71 us with eight -- eight leaves 224 workgroups for 256 CUs)
    if (p.N >= 16384 && p.M > 8) return launch_mt<EPI, 4, 4>(p, stream);        // M = 16: 62.8 -> 55.7 us; M <= 8: +-0
    return launch_mt<EPI, 4, 1>(p, stream);
}

// EPI_PARTIAL geometry: 8 waves x RT row tiles per workgroup, S K-slices, so that a workgroup re-reads X (from L2,
// through the same per-CU load path as its W rows from HBM) once per 16*RT weight rows instead of once per 16: at
// M = 32 / 64 the one-tile kernel moves 2x / 4x as many X bytes as W bytes (wqkv 1.65 / 1.12 TB/s of weights).
// The K-slices restore the workgroup count that the taller workgroups lose.
struct PartialGeom { int rt, splits; };
PartialGeom partial_geom(int N, int K) {
    const int units = K / 256;                 // 8 waves x 32: the K granule of one workgroup
    if (K % 256 != 0 || units == 0) return {0, 0};
    {   // tuning aid: CR_PARTIAL_GEOM="N,K,rt,splits" pins one shape's geometry
        static const char* e = getenv("CR_PARTIAL_GEOM");
        int en, ek, ert, es;
        if (e && sscanf(e, "%d,%d,%d,%d", &en, &ek, &ert, &es) == 4 && en == N && ek == K && ert >= 1 && ert <= 4 &&
            N % (16 * ert) == 0 && es >= 1 && units % es == 0)
            return {ert, es};
    }
    // cost of a launch ~ rounds of 256 workgroups x (bytes per workgroup: (W rows + X rows at a nominal M = 32) / slices,
    // + a fixed per-workgroup share).  Measured at M = 32 / 64 (us): wqkv (3,2) 18.8 / 22.7, (4,8) 19.7 / 29.8,
    // (4,4) 20.3 / 29.0; wo (4,4) 14.2 / 16.8, (2,2) 14.9 / 19.2, (4,8) 14.8 / 20.6; w2 (4,4) 32.4 / 39.4, (4,7) 36.8 / 48.0,
    // (2,2) 40.4 / 53.0 -- against 30.4 / 45.1, 17.3 / 25.3 and 53.8 / 81.9 for the one-tile kernel.
    PartialGeom best{0, 0};
    double best_cost = 1e30;
    for (int rt : {4, 3, 2, 1}) {
        if (N % (16 * rt) != 0) continue;
        const int nb = N / (16 * rt);
        for (int s : {1, 2, 4, 7, 8}) {
            if (units % s != 0) continue;
            const double cost = (double)((nb * s + 255) / 256) * ((rt + 2.0) / s + 0.6) + 0.02 * s;   // + the partial-sum traffic
            if (cost < best_cost - 1e-9) { best_cost = cost; best = {rt, s}; }
        }
    }
    return best;
}

}  // namespace

// The last M % 256 rows (<= 64) of a tiled GEMM whose 256x256 launch would otherwise pay a whole extra round of tiles for them
// (gemm.hip: launch_gemm).  ONE wave per workgroup walks the whole K in ascending 32-steps with the same MFMA and the same
// operand roles as the tiled kernels, so these rows get bit for bit what the tiled kernels would have given them
// (tests/test_gpu_ops.py::test_gemm_tail_rows_take_the_small_kernel); eight k-steps of loads in flight hide the latency that a
// lone 128x128 workgroup per 128 columns would expose K / 64 times over.
bool gemm_tail_supported(int epi, const GemmParams& p) {
    if (p.M > 64 || p.M <= 0 || p.K % 32 != 0 || p.w8 || p.a8) return false;
    if (epi == EPI_LS_RES) return p.res && p.scale;
    if (epi == EPI_RES) return p.res != nullptr;
    if (epi == EPI_SWIGLU) return p.N % 16 == 0;
    return epi == EPI_STORE || epi == EPI_GELU || epi == EPI_F32;
}

int launch_gemm_tail(int epi, const GemmParams& p, hipStream_t stream) {
    switch (epi) {
        case EPI_STORE: return launch_mt8<EPI_STORE, 1, 1, false>(p, stream, 1);
        case EPI_GELU: return launch_mt8<EPI_GELU, 1, 1, false>(p, stream, 1);
        case EPI_LS_RES: return launch_mt8<EPI_LS_RES, 1, 1, false>(p, stream, 1);
        case EPI_RES: return launch_mt8<EPI_RES, 1, 1, false>(p, stream, 1);
        case EPI_SWIGLU: return launch_mt8<EPI_SWIGLU, 1, 1, false>(p, stream, 1);
        case EPI_F32: return launch_mt8<EPI_F32, 1, 1, false>(p, stream, 1);
    }
    return CR_ERR_ARG;
}

int gemm_partial_splits(int N, int K) { return partial_geom(N, K).splits; }

bool gemm_skinny_supported(int epi, const GemmParams& p) {
    if (p.M > 64 || p.K % 128 != 0) return false;
    if (p.w8 && (p.K % 512 != 0 || !p.wscale || (p.ldw & 15))) return false;
    if (epi == EPI_PARTIAL) return p.bias == nullptr && partial_geom(p.N, p.K).splits > 0;
    if (epi == EPI_STORE || epi == EPI_F32) return true;
    if (epi == EPI_RES) return p.res != nullptr;
    if (epi == EPI_SWIGLU) return p.N % 16 == 0;
    return false;
}

int launch_gemm_skinny(int epi, const GemmParams& p, hipStream_t stream) {
    static const bool stream_off = [] { const char* e = getenv("CR_SKINNY_OLD"); return e && e[0] == '1'; }();       // A/B aid
    if (!stream_off && epi != EPI_PARTIAL && gemm_stream_supported(epi, p, 1)) return launch_gemm_stream(epi, p, stream, 1);
    switch (epi) {
        case EPI_STORE: return launch_w<EPI_STORE>(p, stream);
        case EPI_RES: return launch_w<EPI_RES>(p, stream);
        case EPI_SWIGLU: return launch_w<EPI_SWIGLU>(p, stream);
        case EPI_F32: return launch_w<EPI_F32>(p, stream);
        case EPI_PARTIAL: {
            const PartialGeom g = partial_geom(p.N, p.K);
            switch (g.rt) {
                case 4: return launch_mt<EPI_PARTIAL, 8, 4>(p, stream, g.splits);
                case 3: return launch_mt<EPI_PARTIAL, 8, 3>(p, stream, g.splits);
                case 2: return launch_mt<EPI_PARTIAL, 8, 2>(p, stream, g.splits);
                case 1: return launch_mt<EPI_PARTIAL, 8, 1>(p, stream, g.splits);
            }
            return CR_ERR_ARG;
        }
    }
    return CR_ERR_ARG;
}
