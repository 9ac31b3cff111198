// Weight-streaming GEMM for decode, second form:  C[M,N] = epi(X[M,K] . W[N,K]^T),  8 < M <= 64 rows, N > 8192, K % 2048 == 0, bf16 weights
// (w1|w3 and the LM head when more than 16 pages decode together).
//
// gemm_skinny.hip splits K over the waves of a workgroup, so a workgroup's 64 weight rows take the whole X (M x K) through the CU's
// load path beside them: at 64 rows X is as many bytes as W, and the launch moves 2 x W (wqkv / w1|w3 / w2 at 2.0-2.3 x
// their weight bytes, 2.75 TB/s of weights).  Here the waves split N instead:
//   * a workgroup = NW waves (4..8), wave w owns the 16 weight rows n0 + 16 w .. + 15 (one v_mfma_f32_16x16x32_bf16 A tile) and
//     walks the whole K range -- as the FOUR contiguous slices the K-split kernel gives its four waves at these shapes: the
//     accumulator chain restarts at every quarter of K and the quarters are added in ascending order, so every element is the same
//     fp32 sum, bit for bit, as gemm_skinny.hip's, and a row's result does not depend on which of the two kernels its batch takes
//     (tests/test_gpu_ops.py::test_gemm_skinny_swiglu_and_row_independence: 9 rows against 36);
//   * X is shared: a 512-deep chunk of all M rows (<= 64 KiB) arrives ONCE per workgroup by LDS-DMA as 1-KiB fragment sub-tiles
//     (16 rows x 32 k, gemm256's conflict-free image) into one of two buffers and is read back by every wave with ds_read_b128;
//     X through the load path = 1 / NW of the other kernel's;
//   * W goes straight to registers, non-temporal, the next chunk's sixteen 16-byte loads per lane in flight under the current chunk; with p.wsw from the
//     decode-layout copy of the weight (a wave's tile = one contiguous run of KiB blocks: w1|w3 at 64 rows 57.8 -> 43.6 us, the same bits);
//   * NW is chosen per shape so that the grid is a whole number of rounds of the CUs (w1|w3: 7 waves -> 256 workgroups);
//   * epilogues straight from the accumulators (C^T tile: lane = 4 consecutive n of one m): store / +residual / fp32 logits with
//     the reference's rounding points, SwiGLU (rows [8 gate | 8 up] of a tile meet through one lane exchange).
// Measured at 64 rows (scripts/skinny_bench.py): w1|w3 73.5 -> 56.6 us.  The K-sliced partial-sum GEMMs (wqkv, wo, w2) gained 1-3 us
// in this form and lost 1-2 us at one row (a wave walking a whole slice alone keeps fewer loads in flight than eight waves
// splitting it), so they stay with gemm_skinny.hip; so does everything up to 16 rows (w1|w3 at one row: 46 us there, 52 here).
#include <stdlib.h>

#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// 8 e4m3 bytes (two dwords) -> one bf16x8 MFMA fragment (exact: e4m3 is a subset of bf16), as in gemm_skinny.hip
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(unsigned lo, unsigned hi) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    return bf16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// W8 (round 5): the weights are e4m3 bytes with one fp32 scale per row (cr_enable_fp8_decode), walked in 64-deep steps of 16 bytes per lane exactly as
// gemm_skinny.hip's W8 instances walk them -- fragment h (h = 0, 1) of a step pairs the weights' bytes 8h .. 8h + 7 with the X chunk at k + 8h, the quarters of K
// restart the chain, the row scale multiplies the finished fp32 sum -- so a row's bits do not depend on which of the two kernels its batch takes.
template <int EPI, int MT, int NW, bool W8 = false>
__global__ __launch_bounds__(NW * 64) void gemm_stream_kernel(const GemmParams p, const int ks_len, const int fold) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSC = 16;                          // k-steps of 32 per chunk: 512-deep chunks (256-deep ones, a round trip per 256 k: w1|w3 at 64 rows 59.1 us against 56.6)
    constexpr int SUBS = MT * KSC;                   // 1-KiB sub-tiles (16 rows x 32 k) of a chunk
    constexpr int CH = SUBS * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = ((int)blockIdx.x * NW + wave) * 16;
    const int split = blockIdx.y;
    const int k0 = split * ks_len;
    const int chunks = ks_len / (32 * KSC);

    // decode layout (p.wsw == 1): this wave's tile is one contiguous run of KiB blocks, a chunk = 16 KiB
    const int wstep = p.wsw ? 512 : 32;
    const bf16* wp = p.wsw ? p.W + ((int64_t)min(n0 / 16, (p.N + 15) / 16 - 1) * (p.K / 32) + k0 / 32) * 512 + lane * 8
                           : p.W + (int64_t)min(n0 + (lane & 15), p.N - 1) * p.ldw + k0 + (lane >> 4) * 8;
    // e4m3: 64-deep steps of 16 bytes per lane; decode layout = 1 KiB per (tile, 64-deep step), a chunk = 8 KiB
    const int w8step = p.wsw ? 1024 : 64;
    const unsigned char* wp8 = p.wsw ? (const unsigned char*)p.W + ((int64_t)min(n0 / 16, (p.N + 15) / 16 - 1) * (p.K / 64) + k0 / 64) * 1024 + lane * 16
                                     : (const unsigned char*)p.W + (int64_t)min(n0 + (lane & 15), p.N - 1) * p.ldw + k0 + (lane >> 4) * 16;
    // X staging: sub-tile s = (row tile s >> 3, k-step s & 7); wave w brings s = w, w + NW, ...; lane -> row lane >> 2 of the
    // sub-tile, 16-byte chunk (lane & 3) ^ 2 (row >> 3) (the image gemm256.hip reads without bank conflicts)
    const int srow = lane >> 2, schunk = (lane & 3) ^ ((srow >> 3) << 1);
    auto stage = [&](int buf, int c) {
        for (int s = wave; s < SUBS; s += NW) {
            const int t = s / KSC, ks = s % KSC;
            const bf16* src = p.A + (int64_t)min(t * 16 + srow, p.M - 1) * p.lda + k0 + c * (32 * KSC) + ks * 32 + schunk * 8;
            __builtin_amdgcn_global_load_lds(CR_GLB(src), CR_LDS(smem + buf * CH + s * 1024), 16, 0, 0);
        }
    };
    const int lrow = lane & 15;
    const int lane_off = lrow * 64 + (((lane >> 4) ^ ((lrow >> 3) << 1)) * 16);

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (a lane's 16 weight bytes per step are a bf16x8 in the bf16 form and sixteen e4m3 in the W8 form: the same register type holds either)
    constexpr int WR = W8 ? KSC / 2 : KSC;                   // 16-byte weight registers per chunk
    bf16x8 wa[WR], wb[WR];
    auto loadw = [&](bf16x8* w, int c) {
        if (W8) {
#pragma unroll
            for (int s8 = 0; s8 < WR; s8++) w[s8] = __builtin_nontemporal_load((const bf16x8*)(wp8 + (int64_t)(c * WR + s8) * w8step));
        } else {
#pragma unroll
            for (int ks = 0; ks < KSC; ks++) w[ks] = __builtin_nontemporal_load((const bf16x8*)(wp + (c * KSC + ks) * wstep));
        }
    };
    f32x4 tot[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) tot[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // W8: lane (row lrow, quarter g = lane >> 4) of the 64-deep step s8 holds k = 64 s8 + 16 g + 0..15: X chunks 2 (g & 1) and 2 (g & 1) + 1 of the 32-deep
    // sub-tile 2 s8 + (g >> 1)
    const int g4 = lane >> 4;
    const int x8_off0 = (g4 >> 1) * 1024 + lrow * 64 + (((2 * (g4 & 1)) ^ ((lrow >> 3) << 1)) * 16);
    const int x8_off1 = (g4 >> 1) * 1024 + lrow * 64 + (((2 * (g4 & 1) + 1) ^ ((lrow >> 3) << 1)) * 16);
    auto compute = [&](int buf, const bf16x8* w) {
        if (W8) {
#pragma unroll
            for (int s8 = 0; s8 < WR; s8++) {
                const u32x4_t wq = __builtin_bit_cast(u32x4_t, w[s8]);
                const bf16x8 w0 = fp8x8_to_bf16(wq[0], wq[1]), w1 = fp8x8_to_bf16(wq[2], wq[3]);
#pragma unroll
                for (int t = 0; t < MT; t++) {
                    const char* xb = smem + buf * CH + (t * KSC + 2 * s8) * 1024;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, *(const bf16x8*)(xb + x8_off0), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, *(const bf16x8*)(xb + x8_off1), acc[t], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KSC; ks++)
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const bf16x8 xf = *(const bf16x8*)(smem + buf * CH + (t * KSC + ks) * 1024 + lane_off);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], xf, acc[t], 0, 0, 0);
            }
    };
    // quarter q of K is done after chunk (q + 1) * fold - 1: its sum joins the total (0 + s0, + s1, + s2, + s3: gemm_skinny.hip's
    // `s = 0; s += red[w]` over its four waves) and the chain restarts
    int until = fold;
    auto close_slice = [&](int done) {
        if (done != until) return;
        until += fold;
#pragma unroll
        for (int t = 0; t < MT; t++) {
#pragma unroll
            for (int e = 0; e < 4; e++) { tot[t][e] += acc[t][e]; acc[t][e] = 0.f; }
        }
    };
    stage(0, 0);
    loadw(wa, 0);
    __syncthreads();
    for (int c = 0; c < chunks; c += 2) {
        if (c + 1 < chunks) { stage(1, c + 1); loadw(wb, c + 1); }
        compute(0, wa);
        close_slice(c + 1);
        __syncthreads();
        if (c + 1 < chunks) {
            if (c + 2 < chunks) { stage(0, c + 2); loadw(wa, c + 2); }
            compute(1, wb);
            close_slice(c + 2);
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < MT; t++) acc[t] = tot[t];
    if (W8) {                                                 // the row's scale, once, on the finished fp32 sum (gemm_skinny.hip: `s *= p.wscale[n]`)
        float sc[4];
#pragma unroll
        for (int e = 0; e < 4; e++) sc[e] = p.wscale[min(n0 + (lane >> 4) * 4 + e, p.N - 1)];
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[t][e] *= sc[e];
    }

    // ---- epilogue: lane holds n = n0 + 4 (lane >> 4) + 0..3 of row m = 16 t + (lane & 15) ----
    const int g = lane >> 4, gn = n0 + g * 4;
    if (EPI == EPI_SWIGLU) {
        // rows 0..7 of the tile are gates, 8..15 the matching ups: lane (g, m) holds gate 4 g + e for g < 2 and finds its up in lane + 32
        const int gno = n0 / 2 + g * 4;
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int gm = t * 16 + lrow;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float up = __shfl_xor(acc[t][e], 32, 64);
                o[e] = f2bf(rbf(silu(rbf(acc[t][e]))) * rbf(up));
            }
            if (g < 2 && gm < p.M && gno + 4 <= p.N / 2) *(bf16x4*)((bf16*)p.C + (int64_t)gm * p.ldc + gno) = o;
        }
        return;
    }
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; e++) bias[e] = (p.bias && gn + e < p.N) ? bf2f(p.bias[gn + e]) : 0.f;
#pragma unroll
    for (int t = 0; t < MT; t++) {
        const int gm = t * 16 + lrow;
        if (gm >= p.M) continue;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (gn + e >= p.N) continue;
            float x = rbf(acc[t][e] + bias[e]);
            if (EPI == EPI_F32) { ((float*)p.C)[(int64_t)gm * p.ldc + gn + e] = x; continue; }
            if (EPI == EPI_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn + e]) + x;
            ((bf16*)p.C)[(int64_t)gm * p.ldc + gn + e] = f2bf(x);
        }
    }
}

template <int EPI, int MT, int NW, bool W8>
int launch_nw8(const GemmParams& p, hipStream_t stream, int splits) {
    constexpr int LDS = 2 * MT * 16 * 1024;
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)gemm_stream_kernel<EPI, MT, NW, W8>, LDS)) return CR_ERR_HIP;
    const dim3 grid((p.N + 16 * NW - 1) / (16 * NW), splits);
    hipLaunchKernelGGL((gemm_stream_kernel<EPI, MT, NW, W8>), grid, dim3(NW * 64), LDS, stream, p, p.K / splits, p.K / splits / 2048);      // chunks per quarter of K
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

template <int EPI, int MT, int NW>
int launch_nw(const GemmParams& p, hipStream_t stream, int splits) {
    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_F32) {      // the e4m3 form exists for what decode streams through this kernel: w1|w3 and the LM head
        if (p.w8) return launch_nw8<EPI, MT, NW, true>(p, stream, splits);
    }
    if (p.w8) return CR_ERR_ARG;
    return launch_nw8<EPI, MT, NW, false>(p, stream, splits);
}

template <int EPI, int MT>
int launch_mt(const GemmParams& p, hipStream_t stream, int splits, int nw) {
    switch (nw) {
        case 8: return launch_nw<EPI, MT, 8>(p, stream, splits);
        case 7: return launch_nw<EPI, MT, 7>(p, stream, splits);
        case 6: return launch_nw<EPI, MT, 6>(p, stream, splits);
        case 5: return launch_nw<EPI, MT, 5>(p, stream, splits);
        case 4: return launch_nw<EPI, MT, 4>(p, stream, splits);
    }
    return CR_ERR_ARG;
}

template <int EPI>
int launch_e(const GemmParams& p, hipStream_t stream, int splits, int nw) {
    switch ((p.M + 15) / 16) {
        case 1: return launch_mt<EPI, 1>(p, stream, splits, nw);
        case 2: return launch_mt<EPI, 2>(p, stream, splits, nw);
        case 3: return launch_mt<EPI, 3>(p, stream, splits, nw);
        case 4: return launch_mt<EPI, 4>(p, stream, splits, nw);
    }
    return CR_ERR_ARG;
}

}  // namespace

// waves per workgroup for an [N, K / splits] slice grid: the fewest rounds of the CUs times the workgroup's length; ties -> more waves
// (N and K only: the choice never depends on M)
int gemm_stream_waves(int N, int splits) {
    {   // tuning aid
        static const char* e = getenv("CR_STREAM_NW");
        const int v = e ? atoi(e) : 0;
        if (v >= 4 && v <= 8) return v;
    }
    const int cus = cr_device_cus();
    int best = 8;
    long best_cost = 1L << 60;
    for (int nw = 8; nw >= 4; nw--) {
        const long blocks = (long)((N + 16 * nw - 1) / (16 * nw)) * splits;
        const long cost = ((blocks + cus - 1) / cus) * nw;
        if (cost < best_cost) { best_cost = cost; best = nw; }
    }
    return best;
}

bool gemm_stream_supported(int epi, const GemmParams& p, int splits) {
    // exactly the launches gemm_skinny.hip would run with FOUR waves over K and whose quarters are whole 256-deep chunks
    // (9..16 rows too since the weights have their decode layout: w1|w3 at 16 rows 48 -> 41 us; up to 8 rows gemm_decode.hip's kernels run)
    static const int min_rows = [] { const char* e = getenv("CR_STREAM_MIN"); return e ? atoi(e) : 8; }();      // tuning aid
    if (p.a8 || p.M <= min_rows || p.M > 64 || splits != 1 || p.N <= 8192 || p.K % 2048 != 0 || p.wsw > 1) return false;
    if (p.w8 && ((epi != EPI_SWIGLU && epi != EPI_F32) || !p.wscale || (!p.wsw && (p.ldw & 15)) || p.bias)) return false;      // e4m3 weights: the two shapes decode streams here
    if ((p.lda & 7) || (!p.wsw && (p.ldw & 7)) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    if (epi == EPI_STORE || epi == EPI_F32) return true;
    if (epi == EPI_RES) return p.res != nullptr;
    if (epi == EPI_SWIGLU) return p.N % 16 == 0 && (p.ldc & 3) == 0 && ((uintptr_t)p.C & 7) == 0;
    return false;
}

int launch_gemm_stream(int epi, const GemmParams& p, hipStream_t stream, int splits) {
    const int nw = gemm_stream_waves(p.N, splits);
    switch (epi) {
        case EPI_STORE: return launch_e<EPI_STORE>(p, stream, splits, nw);
        case EPI_RES: return launch_e<EPI_RES>(p, stream, splits, nw);
        case EPI_SWIGLU: return launch_e<EPI_SWIGLU>(p, stream, splits, nw);
        case EPI_F32: return launch_e<EPI_F32>(p, stream, splits, nw);
    }
    return CR_ERR_ARG;
}
