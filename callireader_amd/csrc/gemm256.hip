// 256x256x64 bf16 MFMA GEMM for gfx950, 8 waves, LDS-DMA kept in flight across barriers.
//   C = epi(A[M,K] . W[N,K]^T),  K % 128 == 0  (K-tiles are processed in pairs)
//
// Geometry: 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 8 x 4 accumulators of
// v_mfma_f32_16x16x32_bf16 (128 VGPRs).  Per K-tile (64 deep) a wave runs 4 phases of 16 MFMAs, one
// 64 x 32 quadrant (mh, nh) of its sub-tile each: (0,0) (0,1) (1,1) (1,0), so consecutive phases reuse
// either the A or the B fragments already in registers.  The weight fragment is the MFMA's A operand, so a lane's
// accumulator holds four consecutive output COLUMNS of one row (what the epilogue wants).
//
// LDS: 2 K-tile buffers x 4 units x 16 KiB = 128 KiB.  A unit holds what ONE phase's ds_reads consume for ALL
// waves: U0 = A rows of quadrant-row mh=0 (of both wave rows), U1 = B cols of nh=0 (of all four wave columns),
// U2 = B nh=1, U3 = A mh=1.  A unit is 16 sub-tiles of 1 KiB = one MFMA operand fragment (16 rows x 32 k,
// 64-B rows); one global_load_lds_dwordx4 wave-instruction fills one sub-tile, 8 waves x 2 instructions fill
// a unit.  Inside a sub-tile the 16-B chunk of rows 8..15 is XORed with 2 (on the DMA source address and on
// the ds_read_b128 address), which makes the fragment read bank-conflict free.
//
// Schedule.  A phase is two slots, each closed by s_barrier:
//   memory slot: this phase's ds_read_b128s, the two DMA pieces of one unit, s_waitcnt vmcnt(8)
//   matrix slot: s_waitcnt lgkmcnt(0), 16 MFMAs at s_setprio 1
// and waves 4..7 run ONE SLOT BEHIND waves 0..3 (they take an extra barrier at the top of each tile, waves 0..3 one
// at its end).  The two waves that share a SIMD are w and w + 4, so each SIMD always has one wave issuing MFMAs
// while its partner issues LDS reads and DMA: lock-stepped partners left the matrix pipe idle through every
// issue burst (measured on 8192^3 and the prefill shapes: 8-phase lock-step 1.36-1.38 PFLOP/s, + fragments
// prefetched one phase ahead 1.41-1.42, this slot stagger 1.49-1.51; in-kernel stamps: 594-623 clocks per phase
// against 512 of pure MFMA issue).
//   one iteration = K-tiles 2i [even buffer] and 2i+1 [odd buffer], 8 phases:
//   phase   reads (ds_read_b128 -> regs)        MFMA quadrant     DMA issued (2 per wave)
//     1     even U0 (8) + U1 (4)                 (0,0)             odd  U2  of K-tile 2i+1
//     2     even U2 (4)                          (0,1)             odd  U3
//     3     even U3 (8)                          (1,1)             even U0  of K-tile 2i+2
//     4     -                                    (1,0)             even U1
//     5     odd  U0 + U1                         (0,0)             even U2
//     6     odd  U2                              (0,1)             even U3
//     7     odd  U3                              (1,1)             odd  U0  of K-tile 2i+3
//     8     -                                    (1,0)             odd  U1
//   * RAW: a unit is staged 5 or 6 phases before the phase that reads it; vmcnt(8) at the end of every memory slot
//     leaves the four most recent units in flight, so what the NEXT phase reads has landed in every wave before the
//     barrier that precedes those reads (both wave groups: the late group waits one slot later, the early group
//     reads one slot after that).
//   * WAR: a unit is re-staged two phases (or more) after the phase that read it: the reads of both groups were
//     retired by the lgkmcnt(0) of their matrix slots, the later of which ends one full slot before the early
//     group's DMA into the unit.
//   * the main loop never drains vmcnt to 0; barriers are raw s_barrier (a __syncthreads() would drain the DMA).
//   * measured nulls (kept out of the code): a second barrier at the start of each lock-stepped phase (-2..3 %),
//     write-through (sc1) output stores (-1 %), spreading the workgroups' start times over 8-32 us (+-0: the
//     epilogue's stores are bound per CU, not by the chip), skipping the output stores entirely (10 % at K = 1024),
//     evening the memory slots out to 8/4/8/4 reads by fetching the next K-tile's nh = 0 fragment in slots 4 and 8
//     into the idle B register set (-1..5 %), one barrier per phase with the late group's barrier moved between its
//     memory and matrix slot (-5 % at large K).  Per-slot stamps: every slot takes ~300 clocks whatever it holds
//     (0..12 reads), against 256 of MFMA issue: what is left is the fixed cost of a barrier-closed slot.
//
// Persistent: <= 256 workgroups (one per CU) walk the tile list.  The look-ahead of the schedule runs straight
// through a tile boundary: in the last K-tile pair of a tile, phases 3..8 already stage K-tile 0 and U0/U1 of
// K-tile 1 of the workgroup's NEXT tile, so no CU ever sits in a cold-start load burst (measured: with every CU
// starting a tile at once the prologue costs ~5 us per tile at ~11 B/clk/CU).  The epilogue therefore may not touch
// the K buffers: it stages 16-row bf16 slices through a separate 4 KiB per wave.  The look-ahead is drained
// (vmcnt(0)) before the epilogue, which lets slots 1..4 of the next tile skip their counted wait: on the in-order
// counter that wait would otherwise sit behind the tile's output stores.
// In-kernel stamps (-DCR_DIAG_STAMPS, s_memtime into the buffer passed as `scale`): K = 1024 tiles spend 39.1 k
// clocks in the main loop, 0.4 k draining the look-ahead and 6.0 k in the epilogue.  The clock the chip sustains under
// this kernel, measured as d(s_memtime) / d(s_memrealtime) x 100 MHz after 200 back-to-back launches, is 1.82-1.92 GHz
// (K = 8192 lowest), so the matrix peak actually on offer is ~1.9-2.0 PFLOP/s, not the 2.5 of the 2.4 GHz figure.
#include <stdlib.h>

#include "gemm_epilogue.hpp"
#include "diag.hpp"

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int UNIT = 16384;                       // bytes per unit
constexpr int KBUF = 4 * UNIT;                    // bytes per K-tile buffer
constexpr int LDS_MAIN2 = 2 * KBUF;               // 131072
constexpr int LDS_BYTES2 = LDS_MAIN2 + 8 * 4096;  // + one 16x64 fp32 slice per wave = 163840 (all of the CU's LDS)

#define WAIT_VM6() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#include "gemm256_diag.inc"      // WAIT_VM8 / WAIT_COLD / CR_STORE_OUT / KI_VALU and the GM256_* hooks: the product's definitions, and the diagnostic builds' (diag.hpp is the door)
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)   /* lgkmcnt(0); the builtin (unlike inline asm) is seen by the compiler's own wait insertion */

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
// ---- GELU by table (EPI_GELU) ----------------------------------------------------------------------------------------
// The input of the GELU is a bf16 value (the rounded linear output), its result is rounded to bf16: a function of 16 bits.  Every
// workgroup fills a table of it in LDS at its start (gelu_erf on each entry: the table IS the formula, bit for bit) for the inputs
// 2^-17 <= |x| < 2^5 (22 exponents x 128 mantissas per sign, 11 KiB), and the epilogue turns ~17 packed fp32 operations + rcp + exp2
// per pair of elements (the vector pipe's whole throughput for 12 k of fc1's 19 k epilogue clocks) into ~9 integer
// operations and two 2-byte LDS gathers.  A slice that holds a value outside the window (zero, denormal-small, |x| >= 32, inf, NaN:
// a wave-uniform vote on the packed maximum of the slice's magnitudes) is redone with the formula.
// Layout of the wave-private 32 KiB behind the K buffers for this instance: positive table at 0, negative table at 8192 (the sign
// bit lands on address bit 13), ONE 2 KiB staging slice per wave in the space around them (wave 0 at 5632, waves 1..7 from 13824).
constexpr int LUT_LO = (127 - 17) << 7;           // bf16 bits of 2^-17
constexpr int LUT_N = 22 * 128;                   // entries per sign
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;

__device__ __forceinline__ void gelu_lut_fill(char* lut, int tid) {
    for (int i = tid; i < 2 * LUT_N; i += 512) {
        const int sgn = i >= LUT_N, k = i - sgn * LUT_N;
        const unsigned bits = (unsigned)(LUT_LO + k) | ((unsigned)sgn << 15);
        const float y = gelu_erf(__uint_as_float(bits << 16));
        *(unsigned short*)(lut + sgn * 8192 + k * 2) = (unsigned short)(__builtin_bit_cast(unsigned short, f2bf(y)));
    }
}
// packed pair of bf16 inputs -> the byte offsets of their table entries (low / high half); mx accumulates the pair's window offsets
__device__ __forceinline__ unsigned gelu_lut_offsets(unsigned pk, unsigned& mx) {
    const u16x2_t d = __builtin_bit_cast(u16x2_t, pk) - u16x2_t{(unsigned short)LUT_LO, (unsigned short)LUT_LO};
    const unsigned t = __builtin_bit_cast(unsigned, d);
    const u16x2_t m = __builtin_bit_cast(u16x2_t, t & 0x7fff7fffu), mo = __builtin_bit_cast(u16x2_t, mx);
    mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(m, mo));
    const u16x2_t d2 = d << u16x2_t{1, 1};
    return (__builtin_bit_cast(unsigned, d2) & 0x1ffe1ffeu) | ((t & 0x80008000u) >> 2);
}

template <int EPI, bool F8 = false>
__device__ __forceinline__ void stage_slice(const f32x4 (&a)[4], const float (&bias_f)[4][4], const float (&scale_f)[4][4],
                                            bool has_bias, char* buf, int lane, const float (&dq_f)[4][4], float dq_row, const char* lut = nullptr) {
    const int r = lane & 15, g = lane >> 4;
    if (EPI == EPI_GELU && lut) {
        unsigned mx = 0;
        unsigned short glo[4][2], ghi[4][2];                  // all sixteen gathers of the slice go out before the first is waited for
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float x[4] = {a[j][0], a[j][1], a[j][2], a[j][3]};
            if (F8) {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] *= dq_row * dq_f[j][e];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] += bias_f[j][e];           // -0.0f where there is no bias: x + -0 = x, bit for bit
            typedef __attribute__((ext_vector_type(2))) float f32x2_;
            const unsigned pk[2] = {__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{x[0], x[1]}, bf16x2)),   // bf16(acc + bias)
                                    __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{x[2], x[3]}, bf16x2))};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned u = gelu_lut_offsets(pk[h], mx);
                glo[j][h] = *(const unsigned short*)(lut + (u & 0xffffu));
                ghi[j][h] = *(const unsigned short*)(lut + (u >> 16));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
            *(u32x2_*)(buf + r * 128 + (((j * 4 + g) ^ r) << 3)) =
                u32x2_{(unsigned)glo[j][0] | ((unsigned)ghi[j][0] << 16), (unsigned)glo[j][1] | ((unsigned)ghi[j][1] << 16)};
        }
        const bool out = (mx & 0xffffu) >= (unsigned)LUT_N || (mx >> 16) >= (unsigned)LUT_N;
        if (__builtin_amdgcn_ballot_w64(out) == 0) return;        // wave-uniform; otherwise the formula below rewrites the slice
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float x[4] = {a[j][0], a[j][1], a[j][2], a[j][3]};
        if (F8) {                                             // e4m3 operands: the sum of products times the row's and the column's scale
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] *= dq_row * dq_f[j][e];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) x[e] += bias_f[j][e];               // -0.0f where there is no bias: x + -0 = x, bit for bit
        if (EPI == EPI_GELU || EPI == EPI_LS_RES) {          // bf16(acc + bias) is a value of its own before the next op
            round_pair_bf16(x[0], x[1], x[0], x[1]);
            round_pair_bf16(x[2], x[3], x[2], x[3]);
            if (EPI == EPI_GELU) {                            // pairs: packed fp32 math (common.hpp: gelu_erf2), the same bits per element
                const f32x2 g0 = gelu_erf2(f32x2{x[0], x[1]}), g1 = gelu_erf2(f32x2{x[2], x[3]});
                x[0] = g0[0]; x[1] = g0[1]; x[2] = g1[0]; x[3] = g1[1];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = x[e] * scale_f[j][e];
            }
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

template <int EPI, bool F8 = false>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, const f32x4 (&acc)[8][4], char* stg, int row_base, int col0,
                                              int lane, const char* lut = nullptr, const char* pre = nullptr) {
    constexpr bool ADD_ROWS = (EPI == EPI_LS_RES || EPI == EPI_RES);
    const bool has_bias = p.bias != nullptr;
    const bool full_n = col0 + 64 <= p.N;
    // ---- per-column operands in the accumulator layout
    float bias_f[4][4], scale_f[4][4], dq_f[4][4];
    if (F8) {                                                 // column (weight-row) scales; N % 64 == 0 on this path
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x4 w = *(const f32x4*)(p.wscale + min(col0 + j * 16 + (lane >> 4) * 4, p.N - 4));
#pragma unroll
            for (int e = 0; e < 4; e++) dq_f[j][e] = w[e];
        }
    }
    auto dq_row = [&](int mf) { return F8 ? p.ascale[min(row_base + mf * 16 + (lane & 15), p.M - 1)] : 1.0f; };
    if (pre) {
        // bias (bytes 0..127) and LayerScale (128..255) of this wave's 64 columns were brought into its staging area by ONE LDS-DMA instruction at
        // the tile's start (the kernel): loading them here cost every tile a global round trip with the matrix pipe idle -- QKV + bias 0.340 ms
        // against 0.306 without a bias vector at M = 64 575
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int o = (j * 16 + (lane >> 4) * 4) * 2;
            const bf16x4 b = *(const bf16x4*)(pre + o);
#pragma unroll
            for (int e = 0; e < 4; e++) bias_f[j][e] = has_bias ? bf2f(b[e]) : -0.0f;
            if (EPI == EPI_LS_RES) {
                const bf16x4 sc = *(const bf16x4*)(pre + 128 + o);
#pragma unroll
                for (int e = 0; e < 4; e++) scale_f[j][e] = bf2f(sc[e]);
            }
        }
    } else {
        const bool vec = full_n && ((((uintptr_t)p.bias) | ((uintptr_t)p.scale)) & 7) == 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = col0 + j * 16 + (lane >> 4) * 4;
            if (!has_bias) {
#pragma unroll
                for (int e = 0; e < 4; e++) bias_f[j][e] = -0.0f;      // the add stays unconditional (a select per element otherwise)
            } else {
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
            char* buf = stg + (mf & 1) * 2048;
            stage_slice<EPI, F8>(acc[mf], bias_f, scale_f, has_bias, buf, lane, dq_f, dq_row(mf));
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
    auto row_of = [&](int mf, int it) { return row_base + mf * 16 + it * 8 + rin; };
    constexpr bool PRE = ADD_ROWS || EPI == EPI_PATCH;
    // Fast path (wave-uniform): whole 64 columns inside N and 16-byte rows.  Its loop holds no other memory access than the rows'
    // loads and the 16-byte stores, so hipcc counts its waits.  With the element-by-element fallback as a branch of the same loop
    // it put s_waitcnt vmcnt(0) in front of every store: each slice then waited for the rows requested for the NEXT slices -- an
    // HBM round trip per slice (in-kernel stamps, M = 64 575, N = K = 1024: the residual epilogue 22.6 k clocks against 6.9 k for a
    // plain store) whatever the look-ahead.
    const bool fast = full_n && (EPI == EPI_F32 ? (p.ldc & 3) == 0 : (p.ldc & 7) == 0) && (!PRE || ((p.ldr & 7) == 0 && ((uintptr_t)p.res & 15) == 0));
    if (fast && PRE && !F8 && row_base + 128 <= p.M && (EPI != EPI_PATCH || (p.group & 7) == 0)) {
        // Rows to add (residual stream, position embedding), interior tiles.  The counter loads and stores share only orders loads
        // among loads and stores among stores, so with both in flight hipcc has to wait with vmcnt(0): the one-slice-ahead form of
        // this loop (below; still used by the last, ragged row of tiles) waited an HBM round trip per slice -- 22.6 k clocks against
        // 6.9 k for a plain store (in-kernel stamps at M = 64 575, N = K = 1024), the same with three slices ahead.  Here the rows of
        // FOUR slices are requested together (one wave-uniform base per 8 rows + ONE per-lane offset: no address registers), the
        // sums replace them in their registers, the first half's stores and the second half's requests go out back to back: two
        // round trips per sub-tile instead of eight.
        const uint32_t lane_off = (uint32_t)(((int64_t)rin * p.ldr + gn) * 2);
        auto rows_at = [&](int mf, int it) -> const char* {
            const int gm0 = row_base + mf * 16 + it * 8;                       // wave-uniform
            const int64_t rrow = EPI == EPI_PATCH ? 1 + gm0 % p.group : gm0;    // (EPI_PATCH: 8 | group is this path's condition, so a piece never wraps)
            return (const char*)p.res + rrow * p.ldr * 2;
        };
        bf16x8 rr[4][2];
        auto request = [&](int h) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int it = 0; it < 2; it++) rr[q][it] = *(const bf16x8*)(rows_at(4 * h + q, it) + lane_off);
        };
        auto send = [&](int h) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int gm = row_of(4 * h + q, it);
                    int64_t orow = gm;
                    if (EPI == EPI_PATCH) { const int t = gm / p.group; orow = (int64_t)t * (p.group + 1) + 1 + (gm - t * p.group); }
                    __builtin_nontemporal_store(rr[q][it], (bf16x8*)((bf16*)p.C + orow * p.ldc + gn));
                }
        };
        request(0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int mf = 4 * h + q;
                char* buf = stg + (mf & 1) * 2048;
                stage_slice<EPI, F8>(acc[mf], bias_f, scale_f, has_bias, buf, lane, dq_f, dq_row(mf), lut);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const bf16x8 v = read_chunk(buf, it * 8 + rin, c);
#pragma unroll
                    for (int e = 0; e < 8; e++) rr[q][it][e] = f2bf(bf2f(rr[q][it][e]) + bf2f(v[e]));
                }
                __builtin_amdgcn_wave_barrier();
            }
            send(h);
            if (h == 0) request(1);                          // behind the first half's stores: both kinds drain together, once
        }
        return;
    }
    if (fast) {
        constexpr int RPF = 1;                              // (e4m3 instance: no registers for more than one slice of rows ahead)
        bf16x8 rring[RPF + 1][2];
        auto load_rows = [&](int mf, bf16x8 (&dst)[2]) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int gm = min(row_of(mf, it), p.M - 1);
                if (EPI == EPI_PATCH) dst[it] = *(const bf16x8*)(p.res + (int64_t)(1 + gm % p.group) * p.ldr + gn);
                else dst[it] = *(const bf16x8*)(p.res + (int64_t)gm * p.ldr + gn);
            }
        };
        if (PRE) {
#pragma unroll
            for (int q = 0; q < RPF; q++) load_rows(q, rring[q]);
        }
#pragma unroll
        for (int mf = 0; mf < 8; mf++) {
            char* buf = stg + (EPI == EPI_GELU ? 0 : (mf & 1) * 2048);      // the GELU instance's tables leave room for one slice per wave
            stage_slice<EPI, F8>(acc[mf], bias_f, scale_f, has_bias, buf, lane, dq_f, dq_row(mf), lut);
            __builtin_amdgcn_wave_barrier();
            if (PRE && mf + RPF < 8) load_rows(mf + RPF, rring[(mf + RPF) % (RPF + 1)]);
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int gm = row_of(mf, it);
                const bf16x8 v = read_chunk(buf, it * 8 + rin, c);
                int64_t orow = gm;
                if (EPI == EPI_PATCH) { const int t = gm / p.group; orow = (int64_t)t * (p.group + 1) + 1 + (gm - t * p.group); }
                if (EPI == EPI_F32) {
                    float* cp = (float*)p.C + orow * p.ldc + gn;
                    if (gm < p.M) {
                        *(f32x4*)cp = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
                        *(f32x4*)(cp + 4) = f32x4{bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7])};
                    }
                } else if (PRE) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(rring[mf % (RPF + 1)][it][e]) + bf2f(v[e]));
                    if (gm < p.M) __builtin_nontemporal_store(o, (bf16x8*)((bf16*)p.C + orow * p.ldc + gn));
                } else {
                    if (gm < p.M) CR_STORE_OUT(v, (bf16x8*)((bf16*)p.C + orow * p.ldc + gn));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // ragged N or unaligned rows: element by element
#pragma unroll
    for (int mf = 0; mf < 8; mf++) {
        char* buf = stg + (EPI == EPI_GELU ? 0 : (mf & 1) * 2048);
        stage_slice<EPI, F8>(acc[mf], bias_f, scale_f, has_bias, buf, lane, dq_f, dq_row(mf), lut);
        __builtin_amdgcn_wave_barrier();
        for (int it = 0; it < 2; it++) {
            const int gm = row_of(mf, it);
            const bf16x8 v = read_chunk(buf, it * 8 + rin, c);
            if (gm >= p.M || gn >= p.N) continue;
            int64_t orow = gm;
            if (EPI == EPI_PATCH) { const int t = gm / p.group; orow = (int64_t)t * (p.group + 1) + 1 + (gm - t * p.group); }
            for (int e = 0; e < 8 && gn + e < p.N; e++) {
                float x = bf2f(v[e]);
                if (ADD_ROWS) x = bf2f(p.res[(int64_t)gm * p.ldr + gn + e]) + x;
                if (EPI == EPI_PATCH) x = x + bf2f(p.res[(int64_t)(1 + gm % p.group) * p.ldr + gn + e]);
                if (EPI == EPI_F32) ((float*)p.C)[orow * p.ldc + gn + e] = x;
                else ((bf16*)p.C)[orow * p.ldc + gn + e] = f2bf(x);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// EPI_GELU_Q8 (e4m3 instance only): fc1's output leaves as e4m3 bytes, row m divided by c8scale[m] -- a bound the LayerNorm in front
// of fc1 derived from its row's norm (norm.hpp: next_scale), so no pass over the finished row is needed.  Per 16-row slice a lane
// packs its 4 consecutive columns into one dword of a [16 rows][64 bytes] image (the dword's 16-byte group XORed with row >> 2:
// the 64 lanes of a ds_write_b32 hit 64 banks), the slice is read back as 16-byte chunks and stored as 64-byte row segments.
__device__ __forceinline__ void epilogue_gelu_q8(const GemmParams& p, const f32x4 (&acc)[8][4], char* stg, int row_base, int col0, int lane) {
    const bool has_bias = p.bias != nullptr;
    const int r = lane & 15, g = lane >> 4;
    float bias_f[4][4], dq_f[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = min(col0 + j * 16 + g * 4, p.N - 4);
        const f32x4 w = *(const f32x4*)(p.wscale + n);
#pragma unroll
        for (int e = 0; e < 4; e++) { dq_f[j][e] = w[e]; bias_f[j][e] = has_bias ? bf2f(p.bias[n + e]) : 0.f; }
    }
    const int rr = lane >> 2, cc = lane & 3;
#pragma unroll
    for (int mf = 0; mf < 8; mf++) {
        char* buf = stg + (mf & 1) * 1024;
        const int m = min(row_base + mf * 16 + r, p.M - 1);
        const float dq_row = p.ascale[m], inv_c = 1.0f / p.c8scale[m];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] = acc[mf][j][e] * (dq_row * dq_f[j][e]) + bias_f[j][e];
            round_pair_bf16(x[0], x[1], x[0], x[1]);
            round_pair_bf16(x[2], x[3], x[2], x[3]);
#pragma unroll
            for (int e = 0; e < 4; e++) x[e] = rbf(gelu_erf(x[e])) * inv_c;
            unsigned d = 0;
            d = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], d, false);
            d = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], d, true);
            *(unsigned*)(buf + r * 64 + (((j ^ (r >> 2)) << 4) | (g << 2))) = d;
        }
        __builtin_amdgcn_wave_barrier();
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 v = *(const u32x4*)(buf + rr * 64 + ((cc ^ (rr >> 2)) << 4));
        const int gm = row_base + mf * 16 + rr, gn = col0 + cc * 16;
        if (gm < p.M && gn + 16 <= p.N) __builtin_nontemporal_store(v, (u32x4*)((unsigned char*)p.C + (int64_t)gm * p.ldc + gn));
        __builtin_amdgcn_wave_barrier();
    }
}

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
// F8 fragments live in the 8-register operand tuples of v_mfma_scale_f32_16x16x128_f8f6f4 from the start: the ds_read_b128 of
// sub-tile ksub lands in registers 4*ksub .. 4*ksub+3.  A lane's 32 k are thus the 16 of its ksub-0 chunk followed by the 16
// of its ksub-1 chunk -- the same k for both operands, which is all a dot product needs.
__device__ __forceinline__ i32x8 f8_put(const i32x8& v, int ks, const bf16x8& frag) {
    const i32x4 x = __builtin_bit_cast(i32x4, frag);
    const i32x8 w = __builtin_shufflevector(x, x, 0, 1, 2, 3, 0, 1, 2, 3);
    return ks == 0 ? __builtin_shufflevector(w, v, 0, 1, 2, 3, 12, 13, 14, 15) : __builtin_shufflevector(v, w, 0, 1, 2, 3, 8, 9, 10, 11);
}

// F8: both operands are e4m3 bytes.  The byte images in memory and in LDS, the DMA, the fragment reads and the schedule are
// those of the bf16 kernel on a matrix of K / 2 "bf16 columns" (the launcher halves K and the leading dimensions); a K-tile
// of 128 bytes per row is 128 k instead of 64, and a matrix slot issues 8 v_mfma_scale_f32_16x16x128_f8f6f4 (unit block
// scales, 32 cycles each) where the bf16 kernel issues 16 v_mfma_f32_16x16x32_bf16 (16 cycles each): the same slot length at
// twice the k.  The per-row / per-column fp32 scales are applied to the finished sum in the epilogue.
template <int EPI, bool F8 = false, bool BIG = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int ntm = (p.M + BM2 - 1) / BM2, ntn = (p.N + BN2 - 1) / BN2;
    const int ntiles = ntm * ntn;
    const int nk = p.K / BK2;

    // tile list index -> (m0, n0): bijective XCD remap (workgroup b and b + 8k share an XCD), then 8-row super-groups
    auto tile_origin = [&](int orig, int& m0, int& n0) {
        const int xcd = orig & 7, q = ntiles >> 3, r = ntiles & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        constexpr int GM = GM256_TILE_GM;                     // 8 (gemm256_diag.inc: CR_TILE_GM for scripts/traffic_clock.py)
        const int per_group = GM * ntn;
        const int grp = pid / per_group;
        const int first_m = grp * GM;
        const int gsz = min(ntm - first_m, GM);
        const int in_g = pid - grp * per_group;
        m0 = (first_m + in_g % gsz) * BM2;
        n0 = (in_g / gsz) * BN2;
    };

    // DMA sources: this wave fills sub-tiles s = 2*wave + e (e = 0,1) of every unit
    //   A units: s -> (wave row s>>3, fragment (s>>1)&3, ksub s&1);  B units: s -> (wave col s>>2, fragment (s>>1)&1, ksub s&1)
    //   lane -> row lane>>2 of the sub-tile, 16-B chunk (lane&3) ^ 2*(row>>3)
    // kept as 32-bit byte offsets from the (uniform) operand base: operands are < 4 GiB, and the fragment double
    // buffer below needs the registers
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((srow >> 3) << 1);
    uint32_t qA[2];           // [mh]   offsets used by phases 2..8 (may already be the NEXT tile's); piece e = 1 is 64 B on
    uint32_t qB[2];           // [nh]
    auto make_ptrs = [&](int m0, int n0) {
        // sub-tiles 2*wave and 2*wave + 1 are the two k-halves (ksub 0 / 1) of the same 16 rows
        int srow = lane >> 2, schunk = (lane & 3) ^ ((srow >> 3) << 1);
        if (F8) {                                             // the e4m3 instance has no register to keep these across a tile: derive them again
            int l = tid;
            asm volatile("" : "+v"(l));
            l &= 63;
            srow = l >> 2; schunk = (l & 3) ^ ((srow >> 3) << 1);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int arow = m0 + (wave >> 2) * 128 + h * 64 + (wave & 3) * 16 + srow;
            const int brow = n0 + (wave >> 1) * 64 + h * 32 + (wave & 1) * 16 + srow;
            qA[h] = (uint32_t)(((int64_t)min(arow, p.M - 1) * p.lda + schunk * 8) * 2);
            qB[h] = (uint32_t)(((int64_t)min(brow, p.N - 1) * p.ldw + schunk * 8) * 2);
            GM256_DIAG_STAGE_OFFSETS(h)
        }
    };
    auto dma = [&](const bf16* base, uint32_t o, int kt, int buf, int u) {
        char* dst = smem + buf * KBUF + u * UNIT + (2 * wave) * 1024;
        GM256_DIAG_POISON(dst)
        const char* src = (const char*)base + (int64_t)kt * (BK2 * 2);
        GM256_DIAG_ALT_DMA(base, src, o, dst, kt)
        __builtin_amdgcn_global_load_lds(CR_GLB(src + o), CR_LDS(dst), 16, 0, 0);
        // the instruction's immediate offset is added to the global AND to the LDS address: M0 is set 64 short
        __builtin_amdgcn_global_load_lds(CR_GLB(src + o), CR_LDS(dst + 1024 - 64), 16, 64, 0);
    };

    const int lrow = lane & 15;
    const int lane_off = lrow * 64 + (((lane >> 4) ^ ((lrow >> 3) << 1)) * 16);
    int a_sub = wm * 8 * 1024 + lane_off;                // + (i*2 + ksub) * 1024
    int b_sub = wn * 4 * 1024 + lane_off;                // + (j*2 + ksub) * 1024
    char* stg = smem + LDS_MAIN2 + wave * 4096;       // epilogue: two 2 KiB bf16 slices per wave
    const char* lut = nullptr;
    if (EPI == EPI_GELU) {                            // ... or the GELU tables and one slice per wave (layout at gelu_lut_fill)
        lut = smem + LDS_MAIN2;
        stg = smem + LDS_MAIN2 + (wave == 0 ? 2 * LUT_N : 8192 + 2 * LUT_N + (wave - 1) * 2048);
    }

    f32x4 acc[8][4];
    bf16x8 ra[4][2];          // A fragments of the current quadrant row: [i][ksub]
    bf16x8 rb[2][2][2];       // B fragments: [nh][j][ksub]
    i32x8 ra8[4] = {};        // F8: the same fragments as 32-byte operands [i], [nh][j]
    i32x8 rb8[2][2] = {};

#define READ_A(buf, unit)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++) \
        { const bf16x8 f_ = *(const bf16x8*)(smem + (buf) * KBUF + (unit) * UNIT + a_sub + (i * 2 + ks) * 1024); \
          if (F8) ra8[i] = f8_put(ra8[i], ks, f_); else ra[i][ks] = f_; }
#define READ_B(buf, unit, nh)                                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++) \
        { const bf16x8 f_ = *(const bf16x8*)(smem + (buf) * KBUF + (unit) * UNIT + b_sub + (j * 2 + ks) * 1024); \
          if (F8) rb8[nh][j] = f8_put(rb8[nh][j], ks, f_); else rb[nh][j][ks] = f_; }
    // a phase = a memory slot (fragment reads of this phase, one unit's two DMA pieces, the counted wait) and a
    // matrix slot (16 MFMAs), each closed by s_barrier.  Waves 4..7 run one slot behind waves 0..3, so on every
    // SIMD one wave's matrix slot runs beside its partner's memory slot.
#define MEM_SLOT_B(READS, DMA, WAIT)      /* 32-MFMA schedule: the reads are retired in front of the slot's barrier */ \
    READS;                                                                                      \
    DMA;                                                                                        \
    KI_VALU();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (WAIT) WAIT_VM8();                                                                       \
    WAIT_LGKM0();                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_barrier();
#define MEM_SLOT(READS, DMA, WAIT)                                                              \
    READS;                                                                                      \
    DMA;                                                                                        \
    KI_VALU();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (WAIT) WAIT_VM8();                                                                       \
    __builtin_amdgcn_s_barrier();
#define MFMA_SLOT(mh, nh)                                                                       \
    WAIT_LGKM0();                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_setprio(1);                                                              \
    if (F8) {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 2; j++) \
            acc[(mh) * 4 + i][(nh) * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(  \
                rb8[nh][j], ra8[i], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0, 0, 0, 0); \
    } else {                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++) _Pragma("unroll") for (int i = 0; i < 4; i++) \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                                       \
                acc[(mh) * 4 + i][(nh) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[nh][j][ks], ra[i][ks], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0); \
    }                                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_barrier();
#define MFMA_QUAD(mh, nh)                                                                       \
    if (F8) {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 2; j++) \
            acc[(mh) * 4 + i][(nh) * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(  \
                rb8[nh][j], ra8[i], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0, 0, 0, 0); \
    } else {                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++) _Pragma("unroll") for (int i = 0; i < 4; i++) \
            _Pragma("unroll") for (int j = 0; j < 2; j++)                                       \
                acc[(mh) * 4 + i][(nh) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[nh][j][ks], ra[i][ks], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0); \
    }
#define MFMA_SLOT2(mh0, nh0, mh1, nh1)                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_setprio(1);                                                              \
    MFMA_QUAD(mh0, nh0)                                                                         \
    MFMA_QUAD(mh1, nh1)                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_barrier();

    GM256_DIAG_KI_DECL
    int t_cur = blockIdx.x;
    if (t_cur >= ntiles) return;
    int m0, n0;
    tile_origin(t_cur, m0, n0);
    make_ptrs(m0, n0);
    // ---- cold start (first tile of this workgroup only): K-tile 0 and U0, U1 of K-tile 1 issued, U0/U1 of K-tile 0 landed
    dma(p.A, qA[0], 0, 0, 0); dma(p.W, qB[0], 0, 0, 1); dma(p.W, qB[1], 0, 0, 2); dma(p.A, qA[1], 0, 0, 3);
    dma(p.A, qA[0], 1, 1, 0); dma(p.W, qB[0], 1, 1, 1);
    if (BIG) dma(p.W, qB[1], 1, 1, 2);                         // (32-MFMA slots: phase 1 of the first pair stages U3 of K-tile 1)
    if (EPI == EPI_GELU) gelu_lut_fill(smem + LDS_MAIN2, tid);      // under the cold-start fills; the barrier below publishes it
    if (EPI == EPI_GELU) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table's ds_writes, explicitly (free under the DMA wait)
    WAIT_COLD();
    __builtin_amdgcn_s_barrier();

    while (true) {
        if (F8) {           // not kept across the epilogue (no register to spare there: a spilled copy would drain vmcnt at the tile's start)
            int l = tid;
            asm volatile("" : "+v"(l));
            const int r16 = l & 15, off = r16 * 64 + ((((l & 63) >> 4) ^ ((r16 >> 3) << 1)) * 16);
            a_sub = wm * 8 * 1024 + off;
            b_sub = wn * 4 * 1024 + off;
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t_next = t_cur + gridDim.x;
        const bool has_next = t_next < ntiles;
        // per-column epilogue operands of this wave's 64 columns -> its staging area, one LDS-DMA instruction (lanes 0..31: bias, 32..63: LayerScale),
        // issued here so that it is the OLDEST vector-memory operation of the tile: the counted waits of the slots below leave the youngest in flight
        // (16-slot instances only: in the 32-slot ones it measured 2-3 % slower -- nine more spilled registers around the tile boundary)
        const bool pre_ok = !BIG && !F8 && EPI != EPI_ARGMAX && EPI != EPI_SWIGLU && (p.N & 63) == 0 && (p.bias || EPI == EPI_LS_RES) &&
                            ((((uintptr_t)p.bias) | ((uintptr_t)(EPI == EPI_LS_RES ? p.scale : nullptr))) & 3) == 0;
        if (pre_ok) {
            const bf16* src = (lane < 32 || EPI != EPI_LS_RES) ? (p.bias ? p.bias : p.scale) : p.scale;
            __builtin_amdgcn_global_load_lds(CR_GLB(src + n0 + wn * 64 + (lane & 31) * 2), CR_LDS(stg), 4, 0, 0);
        }
        GM256_STAMP_BEGIN
        if (wm) __builtin_amdgcn_s_barrier();                  // waves 4..7 start one slot late

        if (BIG) {
        // Round 4: 32-MFMA matrix slots (template parameter BIG; launch_t picks it where it measured faster: N >= 2048 and K <= 8192).  A K-tile is TWO phases -- (0,0)+(0,1) on the A rows of mh = 0, then (1,1)+(1,0) on mh = 1 (the B
        // fragments of both nh stay in registers: the same 64 fragment registers as before) -- so a K-tile costs four barrier-closed slots
        // instead of eight; every slot carried ~45-50 clocks that were neither MFMA issue nor overlapped (in-kernel stamps: 594-623 clocks per
        // 512 of MFMA issue).  The fragment reads are now retired (lgkmcnt(0)) at the END of the memory slot, in front of its barrier: a unit
        // may then be re-staged in the very next slot (both groups' reads of it have been retired before the barrier that slot starts
        // behind), which keeps every stage -> read distance at three phases = six slots:
        //   phase   reads                         MFMAs           DMA (instructions per wave)
        //     1     even U0 U1 U2 (16)            (0,0) (0,1)     odd  U3 of K-tile 2i+1           (2)
        //     2     even U3 (8)                   (1,1) (1,0)     even U0 U1 U2 of K-tile 2i+2     (6)
        //     3     odd  U0 U1 U2 (16)            (0,0) (0,1)     even U3 of K-tile 2i+2           (2)
        //     4     odd  U3 (8)                   (1,1) (1,0)     odd  U0 U1 U2 of K-tile 2i+3     (6)
        //   RAW: what phase p reads was staged in phase p - 3; vmcnt(8) at the end of every memory slot leaves the two most recent phases'
        //   instructions (2 + 6) in flight, so it has landed in every wave before the barrier in front of the reads (late group: one slot later,
        //   early group reads one slot after that).  WAR: above.  The look-ahead through a tile boundary is phases 2..4 of the last pair + phase 1
        //   of the next tile's first pair.
        for (int kt = 0; kt < nk; kt += 2) {
            int k2 = kt + 2;
            const bool wt = kt != 0;                           // first pair of a tile: phases 1-3 read what the drain before the epilogue retired
            MEM_SLOT_B(READ_B(0, 1, 0); READ_B(0, 2, 1); READ_A(0, 0), dma(p.A, qA[1], kt + 1, 1, 3), wt);
            MFMA_SLOT2(0, 0, 0, 1);
            if (k2 >= nk) {                                    // last pair: the look-ahead belongs to the next tile
                if (has_next) { int nm0, nn0; tile_origin(t_next, nm0, nn0); make_ptrs(nm0, nn0); k2 = 0; }
                else k2 = nk - 2;                              // nothing follows: re-load dead units with the bytes they hold
            }
            MEM_SLOT_B(READ_A(0, 3), dma(p.A, qA[0], k2, 0, 0); dma(p.W, qB[0], k2, 0, 1); dma(p.W, qB[1], k2, 0, 2), wt);
            MFMA_SLOT2(1, 1, 1, 0);
            MEM_SLOT_B(READ_B(1, 1, 0); READ_B(1, 2, 1); READ_A(1, 0), dma(p.A, qA[1], k2, 0, 3), true);
            MFMA_SLOT2(0, 0, 0, 1);
            MEM_SLOT_B(READ_A(1, 3), dma(p.A, qA[0], k2 + 1, 1, 0); dma(p.W, qB[0], k2 + 1, 1, 1); dma(p.W, qB[1], k2 + 1, 1, 2), true);
            MFMA_SLOT2(1, 1, 1, 0);
        }
        } else {
        for (int kt = 0; kt < nk; kt += 2) {
            int k2 = kt + 2;                                   // K-tile staged by phases 3..6 (and k2 + 1 by 7, 8, 1', 2')
            // first pair of a tile: everything phases 1..5 read landed before the epilogue of the previous tile (its
            // stores made the compiler drain vmcnt), so slots 1..4 do not wait -- a counted wait there would sit on
            // the in-order counter until the previous tile's output stores are acknowledged
            const bool wt = kt != 0;
            MEM_SLOT(READ_B(0, 1, 0); READ_A(0, 0), dma(p.W, qB[1], kt + 1, 1, 2), wt);
            MFMA_SLOT(0, 0);
            MEM_SLOT(READ_B(0, 2, 1), dma(p.A, qA[1], kt + 1, 1, 3), wt);
            MFMA_SLOT(0, 1);
            if (k2 >= nk) {                                    // last pair: the look-ahead belongs to the next tile
                if (has_next) { int nm0, nn0; tile_origin(t_next, nm0, nn0); make_ptrs(nm0, nn0); k2 = 0; }
                else k2 = nk - 2;                              // nothing follows: re-load dead units with the bytes they hold
            }
            MEM_SLOT(READ_A(0, 3), dma(p.A, qA[0], k2, 0, 0), wt);
            MFMA_SLOT(1, 1);
            MEM_SLOT(, dma(p.W, qB[0], k2, 0, 1), wt);
            MFMA_SLOT(1, 0);
            MEM_SLOT(READ_B(1, 1, 0); READ_A(1, 0), dma(p.W, qB[1], k2, 0, 2), true);
            MFMA_SLOT(0, 0);
            MEM_SLOT(READ_B(1, 2, 1), dma(p.A, qA[1], k2, 0, 3), true);
            MFMA_SLOT(0, 1);
            MEM_SLOT(READ_A(1, 3), dma(p.A, qA[0], k2 + 1, 1, 0), true);
            MFMA_SLOT(1, 1);
            MEM_SLOT(, dma(p.W, qB[0], k2 + 1, 1, 1), true);
            MFMA_SLOT(1, 0);
        }
        }
        GM256_STAMP(1)
        if (!wm) __builtin_amdgcn_s_barrier();                 // waves 0..3 wait out the partners' last matrix slot
        WAIT_VM0();
        GM256_STAMP(2)       // the look-ahead (K-tile 0 and U0, U1 of K-tile 1 of the next tile) has landed: slots 1..4 rely on it

        // ---- epilogue: eight 16-row slices per wave through its private 4 KiB (the K buffers stay untouched) ----
        GM256_EPILOGUE(if (EPI == EPI_GELU_Q8) epilogue_gelu_q8(p, acc, stg, m0 + wm * 128, n0 + wn * 64, lane);
                       else epilogue_tile<EPI, F8>(p, acc, stg, m0 + wm * 128, n0 + wn * 64, lane, lut, pre_ok ? stg : nullptr);)
        GM256_STAMP_END
        if (!has_next) break;
        t_cur = t_next;
        tile_origin(t_cur, m0, n0);
    }
    WAIT_VM0();       // dead re-loads of the final pair must land before the workgroup releases its LDS
}

template <int EPI, bool F8, bool BIG>
int launch_s(const GemmParams& p, hipStream_t stream) {
    const int ntiles = ((p.M + BM2 - 1) / BM2) * ((p.N + BN2 - 1) / BN2);
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)gemm256_kernel<EPI, F8, BIG>, LDS_BYTES2)) return CR_ERR_HIP;
    const int n_cu = cr_device_cus();
    hipLaunchKernelGGL((gemm256_kernel<EPI, F8, BIG>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), LDS_BYTES2, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

// Two schedules of the same sums (bit-identical results): 32-MFMA matrix slots measured 2-5 % faster on the wide outputs at K <= 4096 (ViT QKV
// and fc1, prefill w1|w3, 8192^3), 16-MFMA slots 1-2 % faster on N = 1024 (proj, fc2) and on K = 14336 (w2): profiles/round4/.  CR_GEMM_SLOTS = 16 | 32
// pins one (A/B aid).  The e4m3 instance keeps the 16-slot schedule (its registers are full).
template <int EPI, bool F8 = false>
int launch_t(const GemmParams& p, hipStream_t stream) {
    static const int pin = [] { const char* e = getenv("CR_GEMM_SLOTS"); return e ? atoi(e) : 0; }();
    const int want = p.slots ? p.slots : pin;
    const bool big = !F8 && (want == 32 || (want != 16 && p.N >= 2048 && p.K <= 8192));
    if constexpr (!F8) { if (big) return launch_s<EPI, false, true>(p, stream); }
    return launch_s<EPI, F8, false>(p, stream);
}

}  // namespace

bool gemm256_supported(int epi, const GemmParams& p) {
    if (p.M < 2048 || p.N < 512 || (p.K % 128) != 0) return false;
    // wave-quantisation model (calibrated on the measured shapes): a round of 256x256 tiles on
    // 256 CUs costs 1; a round of 512 co-resident 128x128 tiles covers half the output at ~0.65 of this kernel's rate
    // (ViT shapes: 0.50-0.82 vs 0.78-1.23 PFLOP/s), i.e. 0.75; pick the kernel with the cheaper schedule
    const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const double c256 = (double)((t256 + 255) / 256);
    const double c128 = 0.75 * (double)((t128 + 511) / 512);
    return c256 <= c128;
}

// e4m3 x e4m3 on the matrix cores: the bf16 kernel's byte view of the operands (two fp8 per "bf16 column")
int launch_gemm256_f8(int epi, const GemmParams& p8, hipStream_t stream) {
    if (!p8.a8 || !p8.w8 || !p8.ascale || !p8.wscale || (p8.K % 256) != 0 || (p8.N % 64) != 0 || (p8.lda & 15) || (p8.ldw & 15) ||
        (((uintptr_t)p8.A | (uintptr_t)p8.W | (uintptr_t)p8.wscale) & 15))
        return CR_ERR_ARG;
    GemmParams p = p8;
    p.K = p8.K / 2; p.lda = p8.lda / 2; p.ldw = p8.ldw / 2;
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE, true>(p, stream);
        case EPI_GELU: return launch_t<EPI_GELU, true>(p, stream);
        case EPI_GELU_Q8: return (p.c8scale && (p.ldc & 15) == 0 && (((uintptr_t)p.C) & 15) == 0) ? launch_t<EPI_GELU_Q8, true>(p, stream) : CR_ERR_ARG;
        case EPI_LS_RES: return (p.scale && p.res) ? launch_t<EPI_LS_RES, true>(p, stream) : CR_ERR_ARG;
        case EPI_RES: return p.res ? launch_t<EPI_RES, true>(p, stream) : CR_ERR_ARG;
        case EPI_SWIGLU: return launch_t<EPI_SWIGLU, true>(p, stream);
        case EPI_F32: return launch_t<EPI_F32, true>(p, stream);
    }
    return CR_ERR_ARG;
}

int launch_gemm256(int epi, const GemmParams& p, hipStream_t stream) {
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE>(p, stream);
        case EPI_GELU: return launch_t<EPI_GELU>(p, stream);
        case EPI_LS_RES: return launch_t<EPI_LS_RES>(p, stream);
        case EPI_RES: return launch_t<EPI_RES>(p, stream);
        case EPI_SWIGLU: return launch_t<EPI_SWIGLU>(p, stream);
        case EPI_PATCH: return launch_t<EPI_PATCH>(p, stream);
        case EPI_F32: return launch_t<EPI_F32>(p, stream);
        case EPI_ARGMAX: return launch_t<EPI_ARGMAX>(p, stream);
    }
    return CR_ERR_ARG;
}
