// Small-batch decode (M <= DECODE_FUSED_MAX_ROWS rows): the five linear layers of an InternLM2 decoder layer and the LM head, each with the row-wise
// kernels around it folded in, so that a decoder layer is SIX launches instead of nine (round-3 verdict, item 2).  Used for up to
// DECODE_FUSED_MAX_ROWS (8) rows: a normalised row costs 8 KiB of every workgroup's LDS; up to 8 rows two workgroups still fit a CU (ALIAS / TG
// below), beyond that the weight streams lose their occupancy (wqkv at 16 rows: 31.7 us against 14.6 + 4.7 + the norm's share separately):
//
//   wqkv  : RMSNorm(x) prologue             -> GEMM -> RoPE + split epilogue: q rows, K / V straight into the cache
//                                                       (modeling_internlm2.py:138-143, 359-388, 233-247)
//   attention over the cache, split combine    (attention.hip, unchanged)
//   wo    : X = attention output             -> GEMM -> x += bf16(sum)          (modeling_internlm2.py:655-660)
//   w1|w3 : RMSNorm(x) prologue             -> GEMM -> SwiGLU                   (:261-264)
//   w2    : X = SwiGLU output                -> GEMM -> x += bf16(sum)          (:662-669)
//   LM head: RMSNorm(x) prologue            -> GEMM -> fp32 logits              (:1081-1082)
//
// A row's bits must not depend on the batch it is decoded in (tests/test_gpu_llm.py::test_batched_decode_equals_single), and batches of
// more rows keep the separate kernels -- so every sum here is formed in exactly the order those kernels use:
//   * gemm_skinny.hip's EPI_PARTIAL geometry for wqkv / wo / w2 is S K-slices (workgroups) x 8 waves, slice s / wave w walking the k range
//     (s*8 + w) * K/(8S) upwards, the waves' tiles added in wave order, the slices added in slice order starting from 0.f by the consumer
//     (rope_split_kernel, add_rmsnorm4096_kernel).  Here ONE workgroup of 8 waves owns a 16-row weight tile and every wave keeps S
//     accumulators ("virtual slices"): the same ranges, the same two summation loops -- no fp32 partials in memory.
//   * w1|w3 and the LM head: 4 waves, each a quarter of K, added in wave order (launch_w).
//   * the RMSNorm prologue is the very function the norm kernels run (norm.hpp: rmsnorm_row16), on the residual row in the kernels' own
//     thread layout, written to LDS as the bf16 row the separate kernel would have stored; the GEMM reads its X fragments from there.
//   * RoPE needs column c and c +- 64 of a head together: a weight tile is 8 rows of a head's first half and the 8 rows 64 further on
//     (which rows a workgroup owns changes no output element's sum), and the epilogue is rope_split_kernel's arithmetic.
// Weight bytes are streamed once, non-temporally, straight to registers (two batches of eight 16-B loads per lane in flight); the
// activations come from LDS (prologue forms) or from L2 (wo, w2: at most 16 rows).  cr_finalize keeps every one of these weights a second time in
// the DECODE LAYOUT (decode_swizzle_kernel: the 64 lanes' 16-byte fragments of one 16-row tile and 32-deep k-step side by side, 1 KiB), so a load
// instruction is one contiguous KiB instead of 16 rows x 64 bytes: the four streams of a layer 95 -> 78 us at one row, a step 3.65 -> 3.10 ms.
#include "common.hpp"
#include "decode.hpp"
#include "norm.hpp"

namespace {

constexpr int D4 = 4096, HD = 128, NKV = 8;
constexpr int LIN_FLOATS = 16 * 17;

enum { DEPI_ROPE = 100 };

// PRO: X = RMSNorm(xres) built in LDS and read from there; otherwise X comes as fragment-shaped loads from L2, one batch ahead.
// TG: tile groups per workgroup -- TG * KW waves, group tg = wave / KW works on tile base + tg, all groups share the normalised rows in LDS
//     (w1|w3 and the LM head at 5..8 rows: 64 KiB of rows leave room for two workgroups per CU, so each brings eight waves).
// ALIAS: one tile per workgroup (wqkv): the sums' staging area lies over the normalised rows, which are dead once every wave has left the
//     main loop (one more barrier) -- 66 KiB instead of 84 at eight rows = two workgroups per CU instead of one.
// (Measured and dropped: copying the activation rows of wo / w2 into LDS first -- 9.8 against 9.3 us for wo, 26.8 against 26.5 for w2 at
// one row once two batches of weights are in flight; hoisting the prologue's row loads; a one-wave-per-row prologue.)
// SW: the weights come in the decode layout (decode_swizzle_kernel below): one 1 KiB block per (16-row tile, 32-deep k-step), lane l's 16 bytes at l * 16 --
//     a wave's load instruction reads ONE contiguous KiB instead of 16 rows x 64 bytes (scripts/ubench/persist_stream.hip: the four streams of a layer take
//     68.5 us contiguous against 83.6 us row-wise; the same values in the same registers, so not a bit changes).
// PROW: the RMSNorm prologue with ONE WAVE PER ROW (lane l holds the row's thread slots l, l + 64, l + 128, l + 192 of norm.hip's layout: the same slot sums, the same
//     butterfly over the lane index, the same ((W0 + W1) + W2) + W3 -- the same bits) instead of 256 threads per row: no barrier inside, every row at once (M <= waves), where
//     the 256-thread form takes M / (threads / 256) rounds of two barriers each.  The packed row beside the weight batch in flight needs more than 128 registers (at 128 it
//     spilled 16-24 and lost, profiles/round5/03_*), so these instances are built for two waves per SIMD -- which is all the w1|w3 / LM-head grids ever put on a CU.
template <int EPI, int KW, int VS, int KPR, bool PRO, int TG = 1, bool ALIAS = false, bool SW = false, bool PROW = false>
__global__ __launch_bounds__(TG * KW * 64, (PRO && !PROW) ? 4 : 2) void gemm_decode_kernel(const DecodeGemmParams p) {      // PRO forms: 128 registers, 16 waves per CU
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = VS * KPR;            // 32-deep k-steps per wave and tile
    constexpr int UB = 8;                  // k-steps per register batch; two batches of weight loads are in flight (batches of 16 in the 256-register forms: +-0, round 5)
    static_assert(T % UB == 0 && T / UB >= 2, "whole batches, at least two");
    static_assert(!ALIAS || (PRO && TG == 1), "aliasing is for the one-tile-per-workgroup prologue form");
    constexpr int NB = T / UB;
    constexpr int NTHREADS = TG * KW * 64;
    const int XROW = p.K * 2 + 16;                               // bytes per row in LDS: 16 rows land on 16 different bank groups
    // LDS: [sums: TG x KW x VS tiles of 16 x 17 floats][finished sums: TG tiles][norm scratch: 64 B][PRO: M normalised rows]; ALIAS puts the rows at 0
    constexpr int RED_FLOATS = TG * KW * VS * LIN_FLOATS;
    float* red = (float*)smem;
    float* lin = red + RED_FLOATS;                               // [TG][16][17]
    char* xlds = ALIAS ? smem : (char*)(lin + TG * LIN_FLOATS) + 64;
    float* nred = ALIAS ? (float*)(smem + (size_t)p.M * XROW) : (float*)(lin + TG * LIN_FLOATS);      // 4 floats per 256-thread group of the prologue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave / KW, kp = wave % KW;                    // tile group, K part
    const int kq = (lane >> 4) * 8;
    const int xm = min(lane & 15, p.M - 1);
    const int ntiles = (p.N + 15) / 16;
    const int et = tid & 255, etg = tid >> 8;                    // epilogue: 256 threads per tile group -> (tile row n, batch row m)
    const int en = et & 15, em = et >> 4;
    const bool ethread = etg < TG;                               // (KW = 8: threads 256..511 of a one-group workgroup have no epilogue element)
    red += tg * KW * VS * LIN_FLOATS;

    auto wrow = [&](int tile, int r) -> int {
        if (EPI == DEPI_ROPE) {            // head slot gs = tile / 8: rows 8j..8j+7 of its first half and the 8 rows 64 further on
            const int gs = tile >> 3, j8 = tile & 7;
            return gs * HD + (r < 8 ? 8 * j8 + r : 64 + 8 * j8 + (r - 8));
        }
        return min(tile * 16 + r, p.N - 1);
    };
    // flat k-step i of this wave: slice s = i / KPR, step j = i % KPR of the range (s*KW + kp) * KPR
    auto koff = [&](int i) -> int { return (((i / KPR) * KW + kp) * KPR + (i % KPR)) * 32; };
    constexpr int WSTEP = SW ? 512 : 32;                          // elements between a lane's loads of consecutive k-steps
    const int64_t tile_stride = (int64_t)(p.K / 32) * 512;       // SW: elements per tile
    auto wbase = [&](int t) -> const bf16* {
        return SW ? p.W + (int64_t)t * tile_stride + lane * 8 : p.W + (int64_t)wrow(t, lane & 15) * p.ldw + kq;
    };

    int base = blockIdx.x * TG;                                  // first tile of this workgroup's current set (workgroup-uniform loop)
    if (base >= ntiles) return;
    int tile = min(base + tg, ntiles - 1);                       // a group past the last tile recomputes it and stores nothing
    bf16x8 wr[2][UB];
    auto load_w = [&](bf16x8 (&dst)[UB], const bf16* wp, int b) {
#pragma unroll
        for (int u = 0; u < UB; u++) dst[u] = __builtin_nontemporal_load((const bf16x8*)(wp + (koff(b * UB + u) / 32) * WSTEP));
    };
    const bf16* wp = wbase(tile);
    load_w(wr[0], wp, 0);                                        // travels under the prologue
    if (!PRO) load_w(wr[1], wp, 1);                              // (the RMSNorm prologue needs the registers: its second batch follows it)

    // what the epilogue will need from memory, requested now: its round trips end long before the sums do
    const int etile0 = min(base + etg, ntiles - 1);
    const bool elive0 = ethread && em < p.M && base + etg < ntiles;
    float xpre = 0.f, cs = 0.f, sn = 0.f;
    int seq = 0, pos = 0;
    if (EPI == EPI_RES && elive0) xpre = bf2f(p.xio[(int64_t)em * D4 + etile0 * 16 + en]);
    if (EPI == DEPI_ROPE && elive0) {
        const int j8 = etile0 & 7, c = en < 8 ? 8 * j8 + en : 64 + 8 * j8 + (en - 8);
        seq = p.seqs[em]; pos = p.lens[seq];
        cs = bf2f(p.cosT[(int64_t)pos * HD + c]); sn = bf2f(p.sinT[(int64_t)pos * HD + c]);
    }

    if (PRO && PROW) {
        if (wave < p.M) {                                         // wave-uniform; the row stays packed (32 registers) between the two passes
            const bf16* xr_ = p.xres + (int64_t)wave * D4;
            bf16x8 raw[4][2];
#pragma unroll
            for (int q = 0; q < 4; q++) { raw[q][0] = *(const bf16x8*)(xr_ + (q * 64 + lane) * 16); raw[q][1] = *(const bf16x8*)(xr_ + (q * 64 + lane) * 16 + 8); }
            float wsum[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float x[16];
#pragma unroll
                for (int e = 0; e < 8; e++) { x[e] = bf2f(raw[q][0][e]); x[8 + e] = bf2f(raw[q][1][e]); }
                wsum[q] = wave_sum(rms_sumsq16(x));
            }
            const float tot = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float x[16], y[16];
#pragma unroll
                for (int e = 0; e < 8; e++) { x[e] = bf2f(raw[q][0][e]); x[8 + e] = bf2f(raw[q][1][e]); }
                rms_apply16(x, tot, p.eps, p.gamma + (q * 64 + lane) * 16, y);
                store16((bf16*)(xlds + (size_t)wave * XROW) + (q * 64 + lane) * 16, y);
            }
        }
        __syncthreads();
        load_w(wr[1], wp, 1);
    } else if (PRO) {
        // RMSNorm of the M residual rows into LDS (measured and dropped: one wave per row without barriers -- the row held in registers
        // beside the weight loads spills at 128 registers, re-reading it costs a second round trip: 18.7 against 16.4 us for wqkv at one row)
        constexpr int NG = NTHREADS / 256;                        // 256 threads per row: norm.hip's own layout and function
        const int g = tid >> 8, t = tid & 255;
        for (int m0 = 0; m0 < p.M; m0 += NG) {
            const int m = m0 + g;
            float x[16], y[16];
            load16(p.xres + (int64_t)min(m, p.M - 1) * D4 + t * 16, x);
            rmsnorm_row16(x, p.gamma + t * 16, p.eps, nred + g * 4, t, y);
            if (m < p.M) store16((bf16*)(xlds + (size_t)m * XROW) + t * 16, y);
        }
        __syncthreads();
        load_w(wr[1], wp, 1);
    }
    constexpr bool XL = PRO;
    const bf16* xg = XL ? nullptr : p.X + (int64_t)xm * p.ldx + kq;
    const char* xl = xlds + (size_t)xm * XROW + kq * 2;

    while (true) {
        f32x4 acc[VS];
#pragma unroll
        for (int s = 0; s < VS; s++) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 xr[2][UB];
        if (!XL) {
#pragma unroll
            for (int u = 0; u < UB; u++) xr[0][u] = *(const bf16x8*)(xg + koff(u));
        }
        const int nbase = base + gridDim.x * TG;
        const bool has_next = nbase < ntiles;                    // workgroup-uniform
        const int ntile = min(nbase + tg, ntiles - 1);
        const bf16* wpn = wp;
        if (has_next) wpn = wbase(ntile);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (XL) {
#pragma unroll
                for (int u = 0; u < UB; u++) xr[b & 1][u] = *(const bf16x8*)(xl + koff(b * UB + u) * 2);
            } else if (b + 1 < NB) {                              // fragment-shaped loads from L2, one batch ahead
#pragma unroll
                for (int u = 0; u < UB; u++) xr[(b + 1) & 1][u] = *(const bf16x8*)(xg + koff((b + 1) * UB + u));
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int s = (b * UB + u) / KPR;                 // compile-time after unrolling
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[b & 1][u], xr[b & 1][u], acc[s], 0, 0, 0);
            }
            // the register set just consumed takes batch b + 2 -- of this tile, or the first two of the next one (they travel under the reduction)
            if (b + 2 < NB) {
                load_w(wr[b & 1], wp, b + 2);
            } else if (has_next) {
                load_w(wr[b & 1], wpn, b + 2 - NB);
            }
        }
        wp = wpn;
        if (ALIAS) __syncthreads();                              // every wave is done with the normalised rows: the sums may land on them
        // C^T tile: n = (lane >> 4) * 4 + e, m = lane & 15
#pragma unroll
        for (int s = 0; s < VS; s++)
#pragma unroll
            for (int e = 0; e < 4; e++) red[((kp * VS + s) * 16 + (lane >> 4) * 4 + e) * 17 + (lane & 15)] = acc[s][e];
        __syncthreads();
        const int etile = base + etg;
        const bool elive = ethread && em < p.M && etile < ntiles;
        if (ethread) {
            const float* rg = (const float*)smem + etg * KW * VS * LIN_FLOATS;
            float a = 0.f;
#pragma unroll
            for (int s = 0; s < VS; s++) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < KW; w++) v += rg[((w * VS + s) * 16 + en) * 17 + em];         // the waves' tiles in wave order
                if (VS == 1) a = v; else a += v;                                                  // the slices in slice order, from 0.f
            }
            lin[etg * LIN_FLOATS + en * 17 + em] = a;
        }
        __syncthreads();
        if (elive) {
            const float* lt = lin + etg * LIN_FLOATS;
            if (EPI == EPI_RES) {
                // add_rmsnorm4096_kernel's first half: x = bf16(x + bf16(sum))
                p.xio[(int64_t)em * D4 + etile * 16 + en] = f2bf(rbf(xpre + rbf(lt[en * 17 + em])));
            } else if (EPI == EPI_SWIGLU) {
                if (en < 8) {
                    const float gt = rbf(lt[en * 17 + em]), up = rbf(lt[(8 + en) * 17 + em]);
                    ((bf16*)p.C)[(int64_t)em * p.ldc + etile * 8 + en] = f2bf(rbf(silu(gt)) * up);
                }
            } else if (EPI == EPI_F32) {
                const int gn = etile * 16 + en;
                if (gn < p.N) ((float*)p.C)[(int64_t)em * p.ldc + gn] = rbf(lt[en * 17 + em] + 0.f);
            } else if (EPI == DEPI_ROPE) {
                // rope_split_kernel: the bf16 linear output, rotated (slots 0..4 of a group: 4 q heads and k), split into q rows and the cache
                const int gs = etile >> 3, j8 = etile & 7, grp = gs / 6, slot = gs - grp * 6;
                const int c = en < 8 ? 8 * j8 + en : 64 + 8 * j8 + (en - 8);
                const bf16 xb = f2bf(lt[en * 17 + em]), xpb = f2bf(lt[(en ^ 8) * 17 + em]);
                bf16 y = xb;
                if (slot < 5) {
                    const float sign = c < 64 ? -1.0f : 1.0f;        // rotate_half: (-x2, x1)
                    y = f2bf(rbf(bf2f(xb) * cs) + rbf(sign * bf2f(xpb) * sn));
                }
                bf16* dst;
                if (slot < 4) dst = p.q_out + (int64_t)em * D4 + (grp * 4 + slot) * HD;
                else dst = (slot == 4 ? p.kc : p.vc) + (((int64_t)seq * NKV + grp) * p.max_tokens + pos) * HD;
                dst[c] = y;
            }
        }
        if (!has_next) break;
        base = nbase;
        tile = ntile;
        if (EPI == EPI_RES && ethread && em < p.M && base + etg < ntiles) xpre = bf2f(p.xio[(int64_t)em * D4 + (base + etg) * 16 + en]);
        __syncthreads();                                         // lin / red are rewritten by the next set of tiles
    }
}

template <int KW, int VS, int TG>
constexpr int fixed_lds() { return TG * (KW * VS + 1) * LIN_FLOATS * 4 + 64; }

template <int EPI, int KW, int VS, int KPR, bool PRO, int TG = 1, bool ALIAS = false, bool SW = false, bool PROW = false>
int launch_one(const DecodeGemmParams& p, int grid_cap, hipStream_t st) {
    if (!SW && p.swizzled) return launch_one<EPI, KW, VS, KPR, PRO, TG, ALIAS, true, PROW>(p, grid_cap, st);
    static_assert(!PROW || PRO, "one wave per row is a prologue form");
    if (PROW && p.M > TG * KW) return CR_ERR_ARG;
    const int ntiles = (p.N + 15) / 16;
    const int sets = (ntiles + TG - 1) / TG;
    const int xbytes = PRO ? p.M * (p.K * 2 + 16) : 0;
    const int lds = ALIAS ? (xbytes + 64 > fixed_lds<KW, VS, TG>() ? xbytes + 64 : fixed_lds<KW, VS, TG>()) : fixed_lds<KW, VS, TG>() + xbytes;
    if (lds > 160 * 1024 || (ALIAS && grid_cap != 0)) return CR_ERR_ARG;
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)gemm_decode_kernel<EPI, KW, VS, KPR, PRO, TG, ALIAS, SW, PROW>, 160 * 1024)) return CR_ERR_HIP;
    const int grid = grid_cap > 0 && grid_cap < sets ? grid_cap : sets;
    // the look-ahead across a tile boundary leaves the next tile's batches 0 / 1 in wr[(NB - 2) & 1] / wr[(NB - 1) & 1]: right for an even number of batches only,
    // so an odd one (w2: 7) must never walk a second tile
    if ((VS * KPR / 8) % 2 != 0 && grid < sets) return CR_ERR_ARG;
    hipLaunchKernelGGL((gemm_decode_kernel<EPI, KW, VS, KPR, PRO, TG, ALIAS, SW, PROW>), dim3(grid), dim3(TG * KW * 64), lds, st, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

// Decode layout of a weight matrix: dst[((tile * K/32 + kstep) * 64 + lane) * 8 + e] = W[row(tile, lane & 15)][kstep * 32 + (lane >> 4) * 8 + e], row() being the
// consuming kernel's own tile -> row map (ROPE: gemm_decode_kernel<DEPI_ROPE>'s 8 + 8 rows of a head; rows past N repeat the last one).  One wave per block.
template <bool ROPE>
__global__ __launch_bounds__(256) void decode_swizzle_kernel(const bf16* __restrict__ W, int64_t ldw, int N, int K, bf16* __restrict__ dst) {
    const int ksteps = K / 32;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (N + 15) / 16;
    if (blk >= ntiles * ksteps) return;
    const int tile = (int)(blk / ksteps), ks = (int)(blk % ksteps), r = lane & 15;
    int row;
    if (ROPE) {
        const int gs = tile >> 3, j8 = tile & 7;
        row = gs * HD + (r < 8 ? 8 * j8 + r : 64 + 8 * j8 + (r - 8));
    } else {
        row = min(tile * 16 + r, N - 1);
    }
    *(bf16x8*)(dst + (blk * 64 + lane) * 8) = *(const bf16x8*)(W + (int64_t)row * ldw + ks * 32 + (lane >> 4) * 8);
}

// The same for the e4m3 copies of cr_enable_fp8_decode (gemm_skinny.hip's W8 instances walk K in 64-deep steps, 16 bytes per lane): one 1 KiB block per
// (16-row tile, 64-deep k-step), dst[((tile * K/64 + kstep) * 64 + lane) * 16 + e] = W8[min(tile * 16 + (lane & 15), N - 1)][kstep * 64 + (lane >> 4) * 16 + e].
__global__ __launch_bounds__(256) void decode_swizzle8_kernel(const unsigned char* __restrict__ W, int64_t ldw, int N, int K, unsigned char* __restrict__ dst) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const int ksteps = K / 64;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (N + 15) / 16;
    if (blk >= ntiles * ksteps) return;
    const int tile = (int)(blk / ksteps), ks = (int)(blk % ksteps);
    const int row = min(tile * 16 + (lane & 15), N - 1);
    *(u32x4_t*)(dst + (blk * 64 + lane) * 16) = *(const u32x4_t*)(W + (int64_t)row * ldw + ks * 64 + (lane >> 4) * 16);
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

}  // namespace

static int launch_decode_gemm_(int which, const DecodeGemmParams& p, hipStream_t st);
// the shapes the instances are built for: InternLM2.5-7B (hidden 4096, ff 14336, 32 q / 8 kv heads of 128)
bool decode_fused_supported(int M, int ff) { return M >= 1 && M <= DECODE_FUSED_MAX_ROWS && ff == 14336; }

int launch_decode_gemm(int which, const DecodeGemmParams& p, hipStream_t st) {
    if (p.M < 1 || p.M > DECODE_FUSED_MAX_ROWS || !p.W) return CR_ERR_ARG;
    static const int dflags = env_int("CR_DEC_FLAGS", -1);
    DecodeGemmParams q = p;
    if (dflags >= 0) q.flags = dflags;
    return launch_decode_gemm_(which, q, st);
}

size_t decode_swizzled_bytes(int N, int K) { return (size_t)((N + 15) / 16) * 16 * K * 2; }

int decode_swizzle_weight(int which, const bf16* W, int64_t ldw, int N, int K, bf16* dst, hipStream_t st) {
    if (!W || !dst || N <= 0 || K <= 0 || (K & 31) || (ldw & 7) || (which == DEC_WQKV && (N % HD))) return CR_ERR_ARG;
    const int64_t blocks = (int64_t)((N + 15) / 16) * (K / 32);
    const unsigned grid = (unsigned)((blocks + 3) / 4);
    if (which == DEC_WQKV) hipLaunchKernelGGL(decode_swizzle_kernel<true>, dim3(grid), dim3(256), 0, st, W, ldw, N, K, dst);
    else hipLaunchKernelGGL(decode_swizzle_kernel<false>, dim3(grid), dim3(256), 0, st, W, ldw, N, K, dst);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

size_t decode_swizzled8_bytes(int N, int K) { return (size_t)((N + 15) / 16) * 16 * K; }

int decode_swizzle_weight8(const unsigned char* W, int64_t ldw, int N, int K, unsigned char* dst, hipStream_t st) {
    if (!W || !dst || N <= 0 || K <= 0 || (K & 63) || (ldw & 15)) return CR_ERR_ARG;
    const int64_t blocks = (int64_t)((N + 15) / 16) * (K / 64);
    hipLaunchKernelGGL(decode_swizzle8_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, st, W, ldw, N, K, dst);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

static int launch_decode_gemm_(int which, const DecodeGemmParams& p, hipStream_t st) {
    // persistent grids: w1|w3 has 1792 tiles, the LM head 5785.  Two 256-thread workgroups per CU (one 512-thread one at 5..8 rows) measured best at every row
    // count (w1|w3 at one row, us: 256 -> 43.6, 512 -> 42.5, 640 -> 46.4, 768 -> 43.3, 896 -> 47.9, 1024 -> 42.9, 1792 -> 44.3; at eight rows 49.8 / 51.2 for 512 / 896):
    // every workgroup repeats the RMSNorm prologue, and a grid that is not a multiple of the 256 CUs leaves some of them a workgroup short
    static const int g13 = env_int("CR_DEC_GRID13", 512), ghead = env_int("CR_DEC_GRIDHEAD", 512);
    // one wave per row in the RMSNorm prologue of w1|w3 and the LM head from 3 rows on (round 5, profiles/round5/15_*; us per launch, 256 threads per row -> one wave per row:
    // w1|w3 38.1 -> 38.3 at 2 rows, 40.4 -> 38.0 at 4, 42.0 -> 41.3 at 5, 43.7 -> 41.9 at 8; one row 36.8 -> 38.1: the single round of the 256-thread form is the shorter one there);
    // flags bit 0 forces it, bit 1 forbids it (A/B)
    const bool prow = ((p.M >= 3) || (p.flags & 1)) && !(p.flags & 2);
    switch (which) {
        case DEC_WQKV:
            if (p.N != 6144 || p.K != 4096 || !p.xres || !p.gamma || !p.cosT || !p.sinT || !p.q_out || !p.kc || !p.vc || !p.seqs || !p.lens) return CR_ERR_ARG;
            // (one wave per row here too -- which costs the second workgroup per CU its registers -- measured 15.1-17.2 us against 11.2-15.5 at 1-8 rows: not kept)
            return launch_one<DEPI_ROPE, 8, DEC_SLICES_WQKV, 8, true, 1, true>(p, 0, st);              // one tile per workgroup: sums alias the normalised rows
        case DEC_WO:
            if (p.N != 4096 || p.K != 4096 || !p.X || !p.xio) return CR_ERR_ARG;
            return launch_one<EPI_RES, 8, DEC_SLICES_WO, 4, false>(p, 0, st);
        case DEC_W13:
            if (p.N != 2 * 14336 || p.K != 4096 || !p.xres || !p.gamma || !p.C) return CR_ERR_ARG;
            if (p.M > 4) {
                if (prow) return launch_one<EPI_SWIGLU, 4, 1, 32, true, 2, false, false, true>(p, g13 / 2, st);
                return launch_one<EPI_SWIGLU, 4, 1, 32, true, 2>(p, g13 / 2, st);      // two tiles at a time on shared rows: 16 waves per CU at 5..8 rows
            }
            if (prow) return launch_one<EPI_SWIGLU, 4, 1, 32, true, 1, false, false, true>(p, g13, st);
            return launch_one<EPI_SWIGLU, 4, 1, 32, true>(p, g13, st);
        case DEC_W2:
            if (p.N != 4096 || p.K != 14336 || !p.X || !p.xio) return CR_ERR_ARG;
            return launch_one<EPI_RES, 8, DEC_SLICES_W2, 14, false>(p, 0, st);
        case DEC_HEAD:
            if (p.K != 4096 || !p.xres || !p.gamma || !p.C) return CR_ERR_ARG;
            if (p.M > 4) {
                if (prow) return launch_one<EPI_F32, 4, 1, 32, true, 2, false, false, true>(p, ghead / 2, st);
                return launch_one<EPI_F32, 4, 1, 32, true, 2>(p, ghead / 2, st);
            }
            if (prow) return launch_one<EPI_F32, 4, 1, 32, true, 1, false, false, true>(p, ghead, st);
            return launch_one<EPI_F32, 4, 1, 32, true>(p, ghead, st);
    }
    return CR_ERR_ARG;
}
