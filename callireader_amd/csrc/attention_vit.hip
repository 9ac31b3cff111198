// ViT attention for gfx950, specialised for the InternViT token layout: S = 1 + 128 n tokens (CLS + a whole number of 128-row
// query blocks; 1025 = 1 + 8 * 128 at 448 x 448), d = 64, no mask, q * 2^-3 in bf16, scores rounded to bf16 before the fp32 softmax
// (InternVL/modeling_intern_vit.py:215-232).  Same arithmetic and MFMA data flow as flash_attn_kernel<64, false> (attention.hip:
// swapped K.Q^T with the query on the lane, S accumulators as the P.V operand, V through ds_read_b64_tr_b16, K/V tiles by LDS-DMA),
// but the one odd token no longer costs a ninth, 1-row query block (11 % of the workgroups, three of four waves idle) and a
// seventeenth, 1-key tile per block (6 % of every block's work):
//   * the 1024 PATCH queries form 8 full blocks of 128 and sweep the 1024 PATCH keys as 16 full tiles of 64: no padding, no masks;
//   * the CLS KEY enters each query's softmax as the INITIAL state instead of a tile: s0 = q . k_cls by 32 FMAs per lane, the
//     reference point of the exponentials m = bf16(s0), row sum 1, O = v_cls -- so every tile of the sweep takes the lean softmax
//     path (attention.hip) from the first one on; a row sum that is not finite or >= 2^100 at the END of the sweep (a score ~88 above
//     that point) flags the block, and vit_attn_exact_kernel redoes flagged blocks in the exact rescaling form;
//   * the CLS QUERY is spread over all workgroups of its (tile, head): block b sweeps its key tiles in the rotated order
//     2b+2, ..., 2b+1 (a softmax does not care; it also spreads the eight blocks' K/V fetches over the tiles), so its LAST two tiles
//     are 2b and 2b+1 and are both still in LDS when the sweep ends; each of the four waves then takes one 32-key half of them for the
//     CLS query (4 + 4 MFMAs, exact softmax) and writes an un-normalised partial (m, l, O[64]); vit_cls_combine_kernel merges the
//     4n + 1 partials (the +1: the CLS key itself, by block 0) in index order, like the decode path's split combine.
// Work per (tile, head): 8 x 16.6 tile-times instead of 9 x 17.
// The sweep is bound by the vector pipe (77 % busy against 42 % for the matrix pipe: the reference's own per-score arithmetic), so: 124
// registers = four waves per SIMD, row sums on the matrix pipe, two tiles per loop trip, the softmax part at priority 1, no SLP packing.
#include <stdlib.h>

#include <type_traits>

#include "attention.hpp"
#include "diag.hpp"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_bf16(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_bf16(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }
// round_bf16(x) as an fp32 value in ONE v_cvt_pk_bf16_f32 (0, x): the low half of the result is bf16(0)
__device__ __forceinline__ float rbf1(float x) {
    const f32x2_t v = {0.f, x};
    return __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)));
}

constexpr int D = 64, ROWB = 128, TILE = 64 * ROWB, KS = 4, DB = 2, CPR = 8, RPI = 8;
__device__ __forceinline__ int kswz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int vswz(int r) { return ((r >> 1) & 1) << 2; }

// One 128-query block of one (tile, head).  EXACT = false: the lean sweep, the block's verdict on it (flag), the output, the block's
// share of the CLS query.  EXACT = true: the same block again in the exact rescaling form (vit_attn_exact_kernel, flagged blocks only).
// NW = waves per workgroup (of 128 queries): 4 waves of 32 queries (round 2..4), or -- round 5, CR_VIT_ATTN_NW=2 -- 2 waves of 64: a wave then holds TWO 32-query
// sub-blocks and every K / V fragment it reads from LDS feeds two MFMAs instead of one (the knock-outs of profiles/round5/06_* put the matrix side of this
// kernel -- MFMAs + fragment reads + fills, no softmax -- at 0.281 ms of the launch's 0.346 and the vector side alone at 0.246: the matrix side is the longer
// one, and it is twice its pure MFMA time; what it waits for is LDS).  The same arithmetic per query either way: the same bits.
template <bool EXACT, int NW = 4>
__device__ __forceinline__ void vit_attn_body(const AttnParams& p, const int qb, const int head, const int batch, const int gx) {
    constexpr int QB = 4 / NW;                               // 32-query sub-blocks per wave
    constexpr int IPW = 8 / NW;                              // 8-row staging pieces (K and V each) per wave and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int NT = 2 * gx;                                  // 64-key tiles of patch keys
    const bf16* Qb = p.Q + (int64_t)batch * p.q_bs + (int64_t)head * p.q_hs;
    const bf16* Kc = p.K + (int64_t)batch * p.k_bs + (int64_t)head * p.k_hs;         // row 0 = the CLS key
    const bf16* Vc = p.V + (int64_t)batch * p.v_bs + (int64_t)head * p.v_hs;
    const bf16* Kb = Kc + p.k_rs;                                                     // patch keys, zero-based
    const bf16* Vb = Vc + p.v_rs;
    int qrow[QB];
#pragma unroll
    for (int j = 0; j < QB; j++) qrow[j] = 1 + qb * 128 + (wave * QB + j) * 32 + l31;

    // ---- Q fragment (B operand of K.Q^T): lane (query l31, half hh) holds Q[q][16ks + 8hh .. +7], times 2^-3 in bf16 ----
    auto load_q = [&](int row, bf16x8* dst) {
        const bf16* qp = Qb + (int64_t)row * p.q_rs + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) dst[ks] = *(const bf16x8*)(qp + ks * 16);          // all loads first (attention.hip: the prescale inside the loop
        if (p.q_prescale != 1.0f) {                                                         // made hipcc wait for every load in turn)
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int e = 0; e < 8; e++) dst[ks][e] = f2bf(bf2f(dst[ks][e]) * p.q_prescale);
        }
    };
    // q . k_cls over this lane's 32 of the 64 dimensions, both halves summed
    auto dot_cls_key = [&](const bf16x8* qv) {
        float part = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 k0 = *(const bf16x8*)(Kc + ks * 16 + hh * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) part = fmaf(bf2f(qv[ks][e]), bf2f(k0[e]), part);
        }
        return part + __shfl_xor(part, 32, 64);
    };
    bf16x8 qf[QB][KS];
#pragma unroll
    for (int j = 0; j < QB; j++) load_q(qrow[j], qf[j]);

    // ---- staging: one 32-bit per-lane byte offset for K and one for V (row r = lane / 8 of an 8-row piece, swizzled 16-byte chunk);
    // the piece, the tile and the (batch, head) go into the instruction's scalar base.  LDS-DMA is issued from inline asm (M0 = LDS
    // base, saved and restored inside the statement): hipcc orders every LDS read behind a __builtin_amdgcn_global_load_lds it knows
    // to be in flight (s_waitcnt vmcnt(0) in front of the P.V reads: the NEXT tile's fill sat on this tile's critical path); its
    // landing is waited for by hand, once, in front of the tile's closing barrier.
    const int sr = lane / CPR, scp = lane % CPR;             // row r = 8 * piece + sr: kswz(r) sees the piece's parity (= ii), vswz(r) does not
    unsigned k_voff[IPW];
#pragma unroll
    for (int ii = 0; ii < IPW; ii++) k_voff[ii] = (unsigned)(sr * (int)p.k_rs + ((scp ^ kswz(8 * ii + sr)) * 8)) * 2u;
    const unsigned v_voff = (unsigned)(sr * (int)p.v_rs + ((scp ^ vswz(sr)) * 8)) * 2u;
    auto dma16 = [&](const bf16* sbase, unsigned voff, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // scalar source bases of this wave's two pieces, kept as running pointers: a tile step is one 64-bit scalar add per piece (the
    // 64-bit row * stride products of a per-tile recomputation were ~50 scalar instructions a tile)
    const int64_t k_step = 64 * p.k_rs, v_step = 64 * p.v_rs;               // elements per key tile
    const bf16* ksrc[IPW];
    const bf16* vsrc[IPW];
    auto seek = [&](int kt) {
#pragma unroll
        for (int ii = 0; ii < IPW; ii++) {
            const int64_t row = (int64_t)kt * 64 + (wave * IPW + ii) * RPI;
            ksrc[ii] = Kb + row * p.k_rs;
            vsrc[ii] = Vb + row * p.v_rs;
        }
    };
    auto stage = [&](int buf) {                              // the tile ksrc / vsrc point at, then on to the next one
#pragma unroll
        for (int ii = 0; ii < IPW; ii++) {
            const unsigned dst = smem_base + buf * (2 * TILE) + (wave * IPW + ii) * 1024;
#ifdef CR_POISON     // hazard screen (diagnostic build): the two 1-KiB pieces about to be refilled hold bf16 NaNs until the new bytes land (gemm256.hip explains)
            {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4_p;
                const u32x4_p nan4 = {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};
                char* pz = smem + buf * (2 * TILE) + (wave * IPW + ii) * 1024 + lane * 16;
                *(u32x4_p*)pz = nan4;
                *(u32x4_p*)(pz + TILE) = nan4;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#endif
            dma16(ksrc[ii], k_voff[ii], dst);
            dma16(vsrc[ii], v_voff, dst + TILE);
            ksrc[ii] += k_step;
            vsrc[ii] += v_step;
        }
    };
    f32x16 oacc[QB][DB];
    float m_run[QB], l_run[QB];
    f32x4 lsum[QB];                                         // lean sweep: row sums (softmax32)
#pragma unroll
    for (int j = 0; j < QB; j++) lsum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 sel;
    {
        const int sr_ = lane & 15, sj_ = lane >> 4;
        const bf16 one = f2bf(((sr_ == 0 && !(sj_ & 1)) || (sr_ == 1 && (sj_ & 1))) ? 1.0f : 0.0f);
        sel = bf16x8{one, one, one, one, one, one, one, one};
    }

    // lane-constant pieces of the LDS addresses (attention.hip)
    const int k_lane_off = l31 * ROWB;
    const int k_sw = kswz(l31);
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int v_lane_row = 4 * hh + tq;
    const int v_sw = vswz(v_lane_row);
    const int v_clow = (g & 1) * 2 + (tp >> 1);
    const int v_lane_off = v_lane_row * ROWB + (tp & 1) * 8;

    // S^T block = K[32 keys] . Q^T: 4 MFMAs
    auto qk32 = [&](const char* kbuf, int kb, const bf16x8* qv) {
        f32x16 s;
#ifdef CR_KO_VIT_MFMA   // knock-out (wrong results, cost structure only): no matrix-pipe work and no fragment reads -- what the softmax's vector work takes alone
#pragma unroll
        for (int e = 0; e < 16; e++) s[e] = bf2f(qv[e & 3][e >> 2]) * (float)(kb + 1) + bf2f(*(const bf16*)(kbuf + (lane & 15) * 2));
        return s;
#endif
#pragma unroll
        for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 kf = *(const bf16x8*)(kbuf + kb * 32 * ROWB + k_lane_off + (((2 * ks + hh) ^ k_sw) * 16));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qv[ks], s, 0, 0, 0);
        }
        return s;
    };
    // the same for BOTH of a wave's query sub-blocks off ONE read of every K fragment (NW = 2)
    auto qk32x2 = [&](const char* kbuf, int kb, const bf16x8* qa, const bf16x8* qb_, f32x16& sa, f32x16& sb) {
#pragma unroll
        for (int e = 0; e < 16; e++) { sa[e] = 0.f; sb[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const bf16x8 kf = *(const bf16x8*)(kbuf + kb * 32 * ROWB + k_lane_off + (((2 * ks + hh) ^ k_sw) * 16));
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qa[ks], sa, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qb_[ks], sb, 0, 0, 0);
        }
    };
    // O^T += V^T[32 keys] . P^T: 4 MFMAs; pk = the block's 8 packed bf16 pairs
    auto pv32 = [&](const char* vbuf, int kb, const unsigned* pk, f32x16* o) {
#ifdef CR_KO_VIT_MFMA
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" ::"v"(pk[j]));     // the probabilities are computed and dropped
        return;
#endif
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const u32x4_t pw = {pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int db = 0; db < DB; db++) {
                const int chunk = ((db * 4) ^ v_sw) | v_clow;
                const char* vp = vbuf + (kb * 32 + 16 * s) * ROWB + v_lane_off + chunk * 16;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp + 8 * ROWB));
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
            }
        }
    };

    auto pv32x2 = [&](const char* vbuf, int kb, const unsigned* pka, const unsigned* pkb, f32x16* oa, f32x16* ob) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const u32x4_t pwa = {pka[4 * s], pka[4 * s + 1], pka[4 * s + 2], pka[4 * s + 3]}, pwb = {pkb[4 * s], pkb[4 * s + 1], pkb[4 * s + 2], pkb[4 * s + 3]};
            const bf16x8 pfa = __builtin_bit_cast(bf16x8, pwa), pfb = __builtin_bit_cast(bf16x8, pwb);
#pragma unroll
            for (int db = 0; db < DB; db++) {
                const int chunk = ((db * 4) ^ v_sw) | v_clow;
                const char* vp = vbuf + (kb * 32 + 16 * s) * ROWB + v_lane_off + chunk * 16;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp + 8 * ROWB));
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oa[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfa, oa[db], 0, 0, 0);
                ob[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfb, ob[db], 0, 0, 0);
            }
        }
    };

    // one 32-key block of scores -> P (packed bf16 pairs) and the row-sum update, lean form (attention.hip): the reference point
    // stays where the CLS key put it; per score one v_cvt_pk_bf16_f32 (0, s), one FMA, one v_exp_f32, half an add, half a v_cvt_pk.
    // No branch, no rescale inside the sweep: P is bf16 (fp32's exponent range) and O, l are fp32, so scores up to ~88 above the
    // reference point are still exact relative arithmetic; what lies beyond shows as a non-finite or > 2^100 row sum at the END of the
    // sweep, and the whole block then repeats the sweep in the exact rescaling form (exact_sweep below).  A per-tile check with an
    // in-loop fallback made hipcc copy the 32 accumulators twice per tile (40 v_mov_b64).
    auto softmax32 = [&](const f32x16& sc, unsigned* pk, float m2f, f32x4& lsum) {
#ifdef CR_KO_VIT_SOFTMAX   // knock-out (wrong results, cost structure only): no rounding, FMA, exponential or packing -- what the matrix pipe, the fragment reads and the fills take alone
#pragma unroll
        for (int j = 0; j < 8; j++) pk[j] = __float_as_uint(sc[2 * j]) ^ __float_as_uint(sc[2 * j + 1]);
        if (false)
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(rbf1(sc[2 * j]), LOG2E, -m2f));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(rbf1(sc[2 * j + 1]), LOG2E, -m2f));
            pk[j] = pack_bf16(p0, p1);
        }
        // The row sums come off the matrix pipe: P (the bf16 values P.V multiplies by) as the B operand of two 16x16x32 MFMAs against a
        // 0/1 matrix.  As that instruction reads it, lane l is column l & 15 with the k range 8 (l >> 4) .. + 7, and it holds query
        // l & 31: k ranges 0 and 2 of column c belong to query c, ranges 1 and 3 to query c + 16 -- so row 0 of `sel` is 1 on ranges
        // 0, 2 and row 1 on ranges 1, 3, and D[0][c] / D[1][c] (lanes 0..15, registers 0 / 1) accumulate the sums of queries c / c + 16.
        // 36 v_add_f32 per tile become four MFMAs of 16 cycles: the vector pipe is the busy one here (77 % against the matrix pipe's 42).
        // Numerics (round-3 advisor note): the denominator is therefore the sum of the bf16-ROUNDED probabilities -- the same values the numerator
        // P.V multiplies by, so the quotient is a weighted mean of V rows whose weights sum to exactly 1 (the reference, and the exact re-sweep /
        // the CLS-query partials / attention.hip, sum the fp32 exponentials before rounding: their weights sum to 1 +- 2^-9).  A flagged block is
        // redone AS A WHOLE by the exact kernel, never mixed; both forms sit inside the tests' bound (test_attention_vit_shape: 2 bf16 ulp of
        // the fp32 reference; scripts/attn_bench.py: largest difference between the two kernels 0.0039 on unit-variance data).
#ifndef CR_KO_VIT_MFMA
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const u32x4_t pw = {pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]};
            lsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel, __builtin_bit_cast(bf16x8, pw), lsum, 0, 0, 0);
        }
#endif
    };
    // the same block in the exact form: running maximum, rescale of O and l when it grows (only exact_sweep uses it)
    auto softmax32_exact = [&](const f32x16& sc, unsigned* pk, float& m_run, float& l_run, f32x16* oacc) {
        float mraw = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; e++) mraw = fmaxf(mraw, sc[e]);
        float mloc = rbf1(mraw);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float m2 = m_new * LOG2E;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(rbf1(sc[2 * j]), LOG2E, -m2));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(rbf1(sc[2 * j + 1]), LOG2E, -m2));
            psum += p0 + p1;
            pk[j] = pack_bf16(p0, p1);
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int e = 0; e < 16; e++) oacc[db][e] *= alpha;
    };
    auto init_state = [&]() {                               // the CLS key as the initial softmax state
#pragma unroll
        for (int j = 0; j < QB; j++) {
            m_run[j] = rbf1(dot_cls_key(qf[j]));            // its score, rounded to bf16 like every score: P = exp(0) = 1
            l_run[j] = hh == 0 ? 1.0f : 0.0f;               // row sums are kept per half-wave and added at the end
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const bf16x4 v0 = *(const bf16x4*)(Vc + 32 * db + 8 * g4 + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; e++) oacc[j][db][4 * g4 + e] = bf2f(v0[e]);
                }
        }
    };
    // one sweep over the NT key tiles in the rotated order; ends with tiles 2qb / 2qb + 1 in LDS buffers 0 / 1
    auto sweep = [&](auto exact) {
        int kt = 2 * qb + 2;
        if (kt >= NT) kt -= NT;
        seek(kt);
        stage(0);
        init_state();
        float m2f[QB];
#pragma unroll
        for (int j = 0; j < QB; j++) m2f[j] = m_run[j] * LOG2E;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // two tiles per trip: the buffer a tile sits in is a compile-time constant, so its LDS offsets are instruction immediates
        // (eight v_add_u32 per tile otherwise); NT is even
        auto tile = [&](int i, auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            if (i + 1 < NT) {
                if (++kt == NT) { kt = 0; seek(0); }         // the rotated sweep wraps once
                stage(cur ^ 1);
            }
            const char* kbuf = smem + cur * (2 * TILE);
            const char* vbuf = kbuf + TILE;
            // K.Q^T of both 32-key blocks, then block by block softmax -> P.V: hipcc runs block 0's P.V MFMAs under block 1's softmax.
            // (Measured null, same run: a hand interleave -- every MFMA of S1 and of P.V0 fenced together with a quarter of the other
            // block's softmax.)
            // The vector pipe is the busy unit (softmax): a wave in its softmax / P.V part outranks the waves that are issuing their
            // eight K.Q^T MFMAs, which fit into the issue slots the vector work leaves.  Same run, 255 tiles: no priorities 1.399 ms,
            // this 1.359, the reverse (K.Q^T high) 1.385, one static priority per workgroup parity 1.382.
            if (!decltype(exact)::value) __builtin_amdgcn_s_setprio(0);
            if constexpr (QB == 1) {
                const f32x16 s0 = qk32(kbuf, 0, qf[0]);
                const f32x16 s1 = qk32(kbuf, 1, qf[0]);
                if (!decltype(exact)::value) __builtin_amdgcn_s_setprio(1);
                unsigned pk0[8], pk1[8];
                if (decltype(exact)::value) softmax32_exact(s0, pk0, m_run[0], l_run[0], oacc[0]); else softmax32(s0, pk0, m2f[0], lsum[0]);
                pv32(vbuf, 0, pk0, oacc[0]);
                if (decltype(exact)::value) softmax32_exact(s1, pk1, m_run[0], l_run[0], oacc[0]); else softmax32(s1, pk1, m2f[0], lsum[0]);
                pv32(vbuf, 1, pk1, oacc[0]);
            } else {
                // two sub-blocks A, B per wave: every K / V fragment read feeds both; block by block, as above
                // (K.Q^T of both 32-key blocks first, as in the four-wave form: block 0's P.V MFMAs run under block 1's softmax)
                f32x16 sa[2], sb[2];
                unsigned pka[8], pkb[8];
                qk32x2(kbuf, 0, qf[0], qf[QB - 1], sa[0], sb[0]);
                qk32x2(kbuf, 1, qf[0], qf[QB - 1], sa[1], sb[1]);
                if (!decltype(exact)::value) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    if (decltype(exact)::value) { softmax32_exact(sa[kb], pka, m_run[0], l_run[0], oacc[0]); softmax32_exact(sb[kb], pkb, m_run[QB - 1], l_run[QB - 1], oacc[QB - 1]); }
                    else { softmax32(sa[kb], pka, m2f[0], lsum[0]); softmax32(sb[kb], pkb, m2f[QB - 1], lsum[QB - 1]); }
                    pv32x2(vbuf, kb, pka, pkb, oacc[0], oacc[QB - 1]);
                }
            }
            // the next tile's fill has had this whole tile to land; every wave's pieces must be in before any wave reads them.
            // lgkmcnt(0) as well: hipcc waits for the tile's LAST V fragments behind the barrier (in front of the MFMA that takes them),
            // and behind the barrier another wave's fill of this very buffer may already be landing -- from the vector L1 when a
            // sibling block on the CU has just fetched the same rows: one wave's d >= 32 columns of one tile's share wrong, once in a
            // few launches of 32 640 blocks (scripts/attn_det.py).
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        for (int i = 0; i < NT; i += 2) {
            tile(i, std::integral_constant<int, 0>{});
            tile(i + 1, std::integral_constant<int, 1>{});
        }
    };
    // The lean sweep's verdict (NaN compares false: a non-finite row sum is flagged as well) goes to the block's flag word; flagged
    // blocks are redone by vit_attn_exact_kernel.  (With the exact sweep inside this kernel -- a workgroup-uniform branch behind the
    // vote -- the kernel took 168 registers for code that all but never runs: 3 waves per SIMD; alone the lean form fits 128 and
    // four: 1.478 -> 1.412 ms per 255-tile launch in one process.)
    int* const flags = (int*)(p.part_o + (size_t)p.B * p.H * (4 * gx + 1) * D);
    float l_tot[QB];
    if constexpr (!EXACT) {
        sweep(std::false_type{});
        __builtin_amdgcn_s_setprio(0);
        int mine = 0;
#pragma unroll
        for (int j = 0; j < QB; j++) {   // query q's sum: lane q & 15, register q >> 4; + 1 for the CLS key (exp(0))
            const float v0 = __shfl(lsum[j][0], l31 & 15, 64), v1 = __shfl(lsum[j][1], l31 & 15, 64);
            l_tot[j] = 1.0f + (l31 < 16 ? v0 : v1);
            mine |= !(l_tot[j] < 1.2676506002282294e30f);
        }
        const int bad = __syncthreads_or(mine);
        if (tid == 0) flags[((int64_t)batch * p.H + head) * gx + qb] = bad;
    } else {
        sweep(std::true_type{});
#pragma unroll
        for (int j = 0; j < QB; j++) l_tot[j] = l_run[j] + __shfl_xor(l_run[j], 32, 64);
    }

    // ---- patch queries: normalise and store; lane (query, half) owns d = 32db + 8g4 + 4hh + 0..3 ----
#pragma unroll
    for (int j = 0; j < QB; j++) {
        const float inv = 1.0f / l_tot[j];
        bf16* op = p.O + (int64_t)batch * p.o_bs + (int64_t)qrow[j] * p.o_rs + (int64_t)head * p.o_hs + 4 * hh;
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = f2bf(oacc[j][db][4 * g4 + e] * inv);
                *(bf16x4*)(op + 32 * db + 8 * g4) = o;
            }
    }

    if constexpr (EXACT) return;
    // ---- the CLS query: this wave's 32 keys of the block's last two tiles (still in LDS: tile 2qb in buffer 0, 2qb + 1 in buffer 1)
    load_q(0, qf[0]);                                       // every lane column holds the same query
    const int nsplit = 4 * gx + 1;
    const int64_t prow = ((int64_t)batch * p.H + head) * nsplit;
#pragma unroll
    for (int hf_ = 0; hf_ < QB; hf_++) {                    // four 32-key halves per block, 4 / NW of them per wave
        const int hf = wave * QB + hf_;
        const char* kbuf = smem + (hf >> 1) * (2 * TILE);
        const int kb = hf & 1;
        const f32x16 s = qk32(kbuf, kb, qf[0]);
        float mraw = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; e++) mraw = fmaxf(mraw, s[e]);
        float m = rbf1(mraw);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float m2 = m * LOG2E;
        unsigned pk[8];
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(rbf1(s[2 * j]), LOG2E, -m2));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(rbf1(s[2 * j + 1]), LOG2E, -m2));
            l += p0 + p1;
            pk[j] = pack_bf16(p0, p1);
        }
        l += __shfl_xor(l, 32, 64);
        f32x16 oc[DB];
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int e = 0; e < 16; e++) oc[db][e] = 0.f;
        pv32(kbuf + TILE, kb, pk, oc);
        if (l31 == 0) {
            const int64_t row = prow + qb * 4 + hf;
            if (hh == 0) { p.part_ml[row * 2] = m; p.part_ml[row * 2 + 1] = l; }
            float* po = p.part_o + row * D + 4 * hh;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                    *(f32x4*)(po + 32 * db + 8 * g4) = f32x4{oc[db][4 * g4], oc[db][4 * g4 + 1], oc[db][4 * g4 + 2], oc[db][4 * g4 + 3]};
        }
    }
    if (qb == 0 && wave == 0) {                             // ... and the CLS key itself: a one-key partial, P = 1
        const float s0 = rbf1(dot_cls_key(qf[0]));
        if (l31 == 0) {
            const int64_t row = prow + 4 * gx;
            if (hh == 0) { p.part_ml[row * 2] = s0; p.part_ml[row * 2 + 1] = 1.0f; }
            float* po = p.part_o + row * D + 4 * hh;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const bf16x4 v0 = *(const bf16x4*)(Vc + 32 * db + 8 * g4 + 4 * hh);
                    *(f32x4*)(po + 32 * db + 8 * g4) = f32x4{bf2f(v0[0]), bf2f(v0[1]), bf2f(v0[2]), bf2f(v0[3])};
                }
        }
    }
}

// the two-wave form: 128 threads, 64 queries per wave (CR_VIT_ATTN_NW=2)
__global__ __launch_bounds__(128, 2) void vit_attn2_kernel(const AttnParams p) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * (int)gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * (int)blockIdx.z);
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    vit_attn_body<false, 2>(p, pid % gx, (pid / gx) % gy, pid / (gx * gy), gx);
}
__global__ __launch_bounds__(256, 4) void vit_attn_kernel(const AttnParams p) {
    // XCD-aware order (attention.hip): the query blocks of one (tile, head) read their K/V through ONE L2
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * (int)gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * (int)blockIdx.z);
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    vit_attn_body<false>(p, pid % gx, (pid / gx) % gy, pid / (gx * gy), gx);
}
// One workgroup per tile: redo the (head, query block)s the lean sweep flagged (a score ~88 above the CLS key's: none on real pages).
// The tile's H * nb flag words are fetched in ONE round trip (a loop of dependent one-word loads per (tile, head) cost 47 us per
// 255-tile launch with nothing flagged).
__global__ __launch_bounds__(256) void vit_attn_exact_kernel(const AttnParams p, const int nb) {
    const int batch = blockIdx.x, n = p.H * nb;
    const int* flags = (const int*)(p.part_o + (size_t)p.B * p.H * (4 * nb + 1) * D) + (int64_t)batch * n;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + (int)threadIdx.x;
        if (!__syncthreads_or(j < n ? flags[j] : 0)) continue;
        for (int jj = j0; jj < n && jj < j0 + 256; jj++)
            if (__builtin_amdgcn_readfirstlane(flags[jj])) vit_attn_body<true>(p, jj % nb, jj / nb, batch, nb);
    }
}

// O[cls][d] = sum_s exp(m_s - M) O_s[d] / sum_s exp(m_s - M) l_s over the (tile, head)'s partials, in index order (reproducible).
// One wave per (tile, head): lane s fetches partial s's (m, l) -- ONE round trip for all of them, the maximum and the weights by
// shuffles -- then lane d sums the weighted O rows with every load of the loop independent (a plain loop over a run-time count
// waited for each partial's load in turn: 33 dependent L2 round trips, 15.8 us per launch at 32 tiles).
template <int NS>
__global__ __launch_bounds__(64) void vit_cls_combine_kernel(const AttnParams p) {
    static_assert(NS <= 64, "one lane per partial");
    const int head = blockIdx.x, batch = blockIdx.y, lane = threadIdx.x;
    const int64_t base = ((int64_t)batch * p.H + head) * NS;
    const float m = lane < NS ? p.part_ml[(base + lane) * 2] : -INFINITY;
    const float l = lane < NS ? p.part_ml[(base + lane) * 2 + 1] : 0.f;
    const float M = wave_max(m);
    const float w = lane < NS ? __expf(m - M) : 0.f;
    float L = 0.f, acc = 0.f;
    float o[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) o[s] = p.part_o[(base + s) * D + lane];
#pragma unroll
    for (int s = 0; s < NS; s++) {                           // index order, as before
        const float ws = __shfl(w, s, 64);
        L += ws * __shfl(l, s, 64);
        acc += ws * o[s];
    }
    p.O[(int64_t)batch * p.o_bs + (int64_t)head * p.o_hs + lane] = f2bf(acc / L);
}

// any other partial count (S = 1 + 128 n with n != 8)
__global__ __launch_bounds__(64) void vit_cls_combine_any_kernel(const AttnParams p, int nsplit) {
    const int head = blockIdx.x, batch = blockIdx.y, d = threadIdx.x;
    const int64_t base = ((int64_t)batch * p.H + head) * nsplit;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; s++) M = fmaxf(M, p.part_ml[(base + s) * 2]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < nsplit; s++) {
        const float w = __expf(p.part_ml[(base + s) * 2] - M);
        L += w * p.part_ml[(base + s) * 2 + 1];
        acc += w * p.part_o[(base + s) * D + d];
    }
    p.O[(int64_t)batch * p.o_bs + (int64_t)head * p.o_hs + d] = f2bf(acc / L);
}

}  // namespace

bool vit_attn_supported(const AttnParams& p, int head_dim, bool causal) {
    const char* e = getenv("CR_VIT_ATTN");                // A/B aid: CR_VIT_ATTN=0 keeps the generic kernel (read per call: a bench flips it in one process)
    const bool off = e && e[0] == '0';
    return !off && head_dim == 64 && !causal && p.Sq == p.Sk && p.Sq > 256 && ((p.Sq - 1) & 127) == 0 && p.s_div == 1.0f && p.kv_group == 1 &&
           !p.seq_map && !p.sk_arr && !p.seg && p.part_ml && (p.q_rs & 7) == 0 && (p.k_rs & 7) == 0 && (p.v_rs & 7) == 0 &&
           (p.o_rs & 3) == 0 && (p.q_hs & 7) == 0 && (p.k_hs & 7) == 0 && (p.v_hs & 3) == 0 && (p.o_hs & 3) == 0;
}

// per (tile, head): 4 nb + 1 partials of [m, l] and O[64], then nb flag words
size_t vit_attn_ws_floats(int B, int H, int S) { const int nb = (S - 1) / 128; return (size_t)B * H * ((4 * nb + 1) * (64 + 2) + nb); }

int launch_vit_attn(const AttnParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * TILE;
    static std::atomic<uint64_t> attr_done{0}, attr_done_x{0}, attr_done_2{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)vit_attn_kernel, LDS) || !cr_dyn_lds_once(attr_done_x, (const void*)vit_attn_exact_kernel, LDS) ||
        !cr_dyn_lds_once(attr_done_2, (const void*)vit_attn2_kernel, LDS)) return CR_ERR_HIP;
    const char* e_nw = getenv("CR_VIT_ATTN_NW");                 // read per call (a bench flips it in one process)
    const bool two_waves = e_nw && e_nw[0] == '2';
    const int nb = (p.Sq - 1) / 128;
    AttnParams q = p;
    q.part_o = p.part_ml + (size_t)p.B * p.H * (4 * nb + 1) * 2;       // one allocation: [m, l] pairs, then the O partials
    if (two_waves) hipLaunchKernelGGL(vit_attn2_kernel, dim3(nb, p.H, p.B), dim3(128), LDS, stream, q);
    else hipLaunchKernelGGL(vit_attn_kernel, dim3(nb, p.H, p.B), dim3(256), LDS, stream, q);
    hipLaunchKernelGGL(vit_attn_exact_kernel, dim3(p.B), dim3(256), LDS, stream, q, nb);
    if (nb == 8) hipLaunchKernelGGL(vit_cls_combine_kernel<33>, dim3(p.H, p.B), dim3(64), 0, stream, q);
    else hipLaunchKernelGGL(vit_cls_combine_any_kernel, dim3(p.H, p.B), dim3(64), 0, stream, q, 4 * nb + 1);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}
