// Fused softmax attention for gfx950 (ViT non-causal S=1025 d=64; InternLM2 causal GQA d=128).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.
// K/V tiles of 64 keys arrive by LDS-DMA (global_load_lds_dwordx4) into two LDS buffers.
//   S^T = K . Q^T        v_mfma_f32_32x32x16_bf16, A = K rows from LDS (ds_read_b128, XOR-swizzled
//                        chunks), B = Q fragment held in registers for the whole kernel.
//                        Swapped operands put the QUERY on the lane and the 32 keys of a block in
//                        16 registers x 2 half-waves: row max / row sum are register-local plus one
//                        cross-half shuffle, and the O rescale factor is one scalar per lane.
//   O^T += V^T . P^T     the S^T accumulators, packed to bf16, ARE the B operand (k order permuted
//                        as the MFMA C layout dictates); V^T comes from the row-major V tile through
//                        ds_read_b64_tr_b16 with the same k permutation, so V is never transposed in
//                        memory.
// Rounding points follow the eager reference:
//   ViT  (modeling_intern_vit.py:225-229): q*scale in bf16 (exact, 2^-3), scores rounded to bf16,
//        softmax in fp32 over them;
//   LLM  (modeling_internlm2.py:393-410): scores rounded to bf16, divided by sqrt(d) -> bf16,
//        causal mask, fp32 softmax, probabilities cast to bf16 before .V (we cast the
//        un-normalised exp and divide the fp32 accumulator by the fp32 row sum at the end).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "attention.hpp"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// two floats -> packed bf16 pair (RNE) with one v_cvt_pk_bf16_f32: the vector conversion is what selects the packed
// instruction (two scalar casts compile to two conversions plus shifts); unpacking is integer so LLVM cannot fold the
// rounding away
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_bf16(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_bf16(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// round_bf16(x) as an fp32 value in ONE instruction: v_cvt_pk_bf16_f32 (0, x) -- the low half of the result is
// bf16(0) = 0x0000, so the dword is the rounded value itself (no unpacking shift / mask)
__device__ __forceinline__ float rbf1(float x) {
    const f32x2_t v = {0.f, x};
    return __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)));
}

#ifndef CR_ATTN_LEAN
#define CR_ATTN_LEAN 1
#endif

template <int D> __device__ __forceinline__ int kswz(int r) { return D == 64 ? ((r >> 1) & 7) : (r & 15); }
template <int D> __device__ __forceinline__ int vswz(int r) { return D == 64 ? (((r >> 1) & 1) << 2) : ((r & 3) << 2); }

// Measured nulls at the ViT shape (0.55 ms either way; PMC: waves issue 27 % of their cycles, sit in issue stalls 40 %,
// parked 33 %): three K/V stages with a counted wait instead of two drained ones (-5 %: one workgroup fewer per CU),
// a 128-VGPR budget for four workgroups per CU (+-0), packed fp32 math for the exponent arguments and row sums (+-0),
// 256-row workgroups of 8 waves sharing each K/V tile (half the L2 -> LDS traffic, +-0), s_setprio 1 around the MFMA
// clusters (-3 %), V fragments read ahead of the softmax (-15 %: the 32 registers cost a resident workgroup).  What the
// counters add up to instead: per wave and 64-key tile ~945 cycles of vector issue + 512 of MFMA + ~200 of LDS/scalar
// issue = the ~1690 cycles observed -- on this mix the SIMD's matrix and vector work do not overlap.
template <int D, bool CAUSAL, bool SPLIT = false, bool DIV = false>
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = D * 2;
    constexpr int TILE = 64 * ROWB;
    constexpr int CPR = D / 8;
    constexpr int RPI = 1024 / ROWB;
    constexpr int IPW = 64 / RPI / 4;
    constexpr int KS = D / 16;
    constexpr int DB = D / 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    // XCD-aware order: workgroups b and b + 8 share an XCD (and its L2), so the linear id is remapped bijectively to
    // give every XCD a contiguous run of the (batch, head, x) space: the query blocks (or key splits) of one (batch, head)
    // read their K/V through ONE L2 instead of up to eight
    int bx, head, batch;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int total = gx * gy * (int)gridDim.z;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * (int)blockIdx.z);
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        bx = pid % gx;
        head = (pid / gx) % gy;
        batch = pid / (gx * gy);
    }
    const int qb = SPLIT ? 0 : bx;
    const int split = SPLIT ? bx : 0;
    const int kvh = head / p.kv_group;
    const int slot = p.seq_map ? p.seq_map[batch] : batch;
    const int Sk = p.sk_arr ? p.sk_arr[slot] + p.sk_add : p.Sk;
    const int qi = qb * 128 + wave * 32 + l31;
    const int qi_c = min(qi, p.Sq - 1);

    // ---- Q fragment (B operand of K.Q^T): lane (query l31, half hh) holds Q[q][16ks + 8hh .. +7]
    bf16x8 qf[KS];
    {
        const bf16* qp = p.Q + (int64_t)batch * p.q_bs + (int64_t)qi_c * p.q_rs + (int64_t)head * p.q_hs + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            qf[ks] = *(const bf16x8*)(qp + ks * 16);
            if (p.q_prescale != 1.0f) {
#pragma unroll
                for (int e = 0; e < 8; e++) qf[ks][e] = f2bf(bf2f(qf[ks][e]) * p.q_prescale);
            }
        }
    }

    const bf16* Kb = p.K + (int64_t)slot * p.k_bs + (int64_t)kvh * p.k_hs;
    const bf16* Vb = p.V + (int64_t)slot * p.v_bs + (int64_t)kvh * p.v_hs;

    // staging addresses: per-lane base pointers of tile 0 are computed once; a full tile adds a wave-uniform offset,
    // only the ragged last tile re-derives clamped rows (the loop is VALU-bound: no per-tile 64-bit multiplies)
    const bf16* kbase[IPW];
    const bf16* vbase[IPW];
#pragma unroll
    for (int ii = 0; ii < IPW; ii++) {
        const int r = (wave * IPW + ii) * RPI + lane / CPR;
        const int cp = lane % CPR;
        kbase[ii] = Kb + (int64_t)r * p.k_rs + ((cp ^ kswz<D>(r)) * 8);
        vbase[ii] = Vb + (int64_t)r * p.v_rs + ((cp ^ vswz<D>(r)) * 8);
    }
    auto stage = [&](int buf, int kt) {
        const bool full = kt * 64 + 64 <= Sk;
        const int64_t koff = (int64_t)kt * 64 * p.k_rs, voff = (int64_t)kt * 64 * p.v_rs;      // scalar
#pragma unroll
        for (int ii = 0; ii < IPW; ii++) {
            const bf16 *ks, *vs;
            if (full) {
                ks = kbase[ii] + koff;
                vs = vbase[ii] + voff;
            } else {
                const int r = (wave * IPW + ii) * RPI + lane / CPR;
                const int cp = lane % CPR;
                const int key = min(kt * 64 + r, Sk - 1);
                ks = Kb + (int64_t)key * p.k_rs + ((cp ^ kswz<D>(r)) * 8);
                vs = Vb + (int64_t)key * p.v_rs + ((cp ^ vswz<D>(r)) * 8);
            }
            char* dst = smem + buf * (2 * TILE) + (wave * IPW + ii) * 1024;
            __builtin_amdgcn_global_load_lds(CR_GLB(ks), CR_LDS(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CR_GLB(vs), CR_LDS(dst + TILE), 16, 0, 0);
        }
    };

    int nt = (Sk + 63) / 64;
    if (CAUSAL) {
        const int kmax = p.q_pos0 + min(qb * 128 + 127, p.Sq - 1);
        nt = min(nt, kmax / 64 + 1);
    }
    const int qpos = p.q_pos0 + qi_c;
    const float inv_div = 1.0f / p.s_div;     // reference divides by sqrt(d); x * (1/d) differs from x / d by <= 1 fp32 ulp before the bf16 rounding
    int t_begin = 0;
    if (SPLIT) {
        t_begin = min(split * ATTN_SPLIT_TILES, nt);
        nt = min(nt, t_begin + ATTN_SPLIT_TILES);
    }

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; db++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[db][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // lane-constant pieces of the LDS addresses
    const int k_lane_off = l31 * ROWB;
    const int k_sw = kswz<D>(l31);
    const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int v_lane_row = 4 * hh + tq;
    const int v_sw = vswz<D>(v_lane_row);
    const int v_clow = (g & 1) * 2 + (tp >> 1);
    const int v_lane_off = v_lane_row * ROWB + (tp & 1) * 8;

    if (t_begin < nt) stage(0, t_begin);
    __syncthreads();
    for (int kt = t_begin; kt < nt; kt++) {
        const int cur = (kt - t_begin) & 1;
        if (kt + 1 < nt) stage(cur ^ 1, kt + 1);
        const char* kbuf = smem + cur * (2 * TILE);
        const char* vbuf = kbuf + TILE;
        // a wave whose 32 query rows are all padding (ragged last query block; decode's 4-row blocks) only helps
        // staging (wave-uniform branch)
        if (qb * 128 + wave * 32 < p.Sq) {
        // (a ragged last key tile runs both 32-key halves: the empty one is masked to -inf anyway, and skipping it
        //  cost every tile 16 accumulator-zeroing moves plus two branches on an issue-bound loop)

        // ---- S^T = K . Q^T for two 32-key blocks ----
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) sacc[kb][e] = 0.f;
            {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const bf16x8 kf = *(const bf16x8*)(kbuf + kb * 32 * ROWB + k_lane_off + (((2 * ks + hh) ^ k_sw) * 16));
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kb], 0, 0, 0);
                }
            }
        }
        // ---- scores: reference rounding, masking, online softmax (query = lane) ----
        // A pair of scores is rounded to bf16 by ONE v_cvt_pk_bf16_f32 and stays packed until the exponentials; the
        // row maximum is taken on the raw accumulators (rounding is monotonic, so max(round(s)) = round(max(s))) with
        // v_max3_f32.  Masking only on tiles that need it.
        const bool need_mask = (kt * 64 + 64 > Sk) || (CAUSAL && kt * 64 + 63 > p.q_pos0 + qb * 128 + wave * 32);
        unsigned ppk[2][8];                                   // P as packed bf16 pairs: the PV B-operand, 4 dwords per fragment
        // Lean form for unmasked tiles after the first (no mask, no divisor: the ViT): the reference point of exp(s - m)
        // stays where the first tile put it (<= the true row maximum); P is rounded to bf16, which has fp32's exponent
        // range, and accumulated in fp32, so nothing is lost while P is finite.  Per score: one v_cvt_pk_bf16_f32 (0, s)
        // rounds it in place, one FMA, one v_exp_f32, half a packed add, half a v_cvt_pk -- no v_max3, no unpacking.  A
        // row sum >= 2^60 in any lane (a score ~42 above the reference point, or +inf) falls back to the exact form below.
        bool lean_done = false;
        if (CR_ATTN_LEAN && !CAUSAL && !DIV && kt > t_begin && !need_mask) {
            const float m2f = m_run * LOG2E;
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(rbf1(sacc[kb][2 * i]), LOG2E, -m2f));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(rbf1(sacc[kb][2 * i + 1]), LOG2E, -m2f));
                    q0 += p0; q1 += p1;
                    ppk[kb][i] = pack_bf16(p0, p1);
                }
            const float qs = q0 + q1;
            if (__all(qs < 1.152921504606846976e18f)) { l_run += qs; lean_done = true; }
        }
        if (!lean_done) {
        float mraw = -INFINITY;
        unsigned spk[2][8];
        auto score_pass = [&](auto masked) {
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    float s0 = sacc[kb][e], s1 = sacc[kb][e + 1];
                    if (decltype(masked)::value) {
                        const int key = kt * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        s0 = (key < Sk && (!CAUSAL || key <= qpos)) ? s0 : -INFINITY;
                        s1 = (key + 1 < Sk && (!CAUSAL || key + 1 <= qpos)) ? s1 : -INFINITY;
                    }
                    mraw = fmaxf(fmaxf(mraw, s0), s1);          // one v_max3_f32 per pair
                    unsigned pk = pack_bf16(s0, s1);
                    if (DIV) pk = pack_bf16(lo_bf16(pk) * inv_div, hi_bf16(pk) * inv_div);
                    spk[kb][e >> 1] = pk;
                }
        };
        if (need_mask) score_pass(std::true_type{}); else score_pass(std::false_type{});   // wave-uniform
        float mloc = lo_bf16(pack_bf16(mraw, mraw));
        if (DIV) mloc = lo_bf16(pack_bf16(mloc * inv_div, mloc * inv_div));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // deferred rescale: while no row's maximum grows by more than 8 the reference point m_run stays (P <= e^8 keeps
        // bf16's relative precision) and the O / l rescale is skipped; the vote is wave-uniform
        const bool rescale = !__all(mloc - m_run <= 8.0f);
        const float m_new = rescale ? fmaxf(m_run, mloc) : m_run;
        const float m2 = m_new * LOG2E;                       // exp(s - m) = exp2(s*log2e - m*log2e): one FMA + v_exp_f32
        const float alpha = rescale ? __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E) : 1.0f;
        m_run = m_new;
        // exponentials on pairs: packed fp32 math (v_pk_fma_f32 / v_pk_add_f32) halves the FMA and row-sum issue slots
        f32x2_t psum2 = {0.f, 0.f};
        const f32x2_t l2e = {LOG2E, LOG2E}, negm = {-m2, -m2};
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const f32x2_t sv = {lo_bf16(spk[kb][i]), hi_bf16(spk[kb][i])};
                const f32x2_t arg = __builtin_elementwise_fma(sv, l2e, negm);
                const f32x2_t pv = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                psum2 += pv;
                ppk[kb][i] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2));
            }
        const float psum = psum2[0] + psum2[1];
        l_run = l_run * alpha + psum;
        if (rescale) {
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int e = 0; e < 16; e++) oacc[db][e] *= alpha;
        }
        }

        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                const u32x4 pw = {ppk[kb][4 * s], ppk[kb][4 * s + 1], ppk[kb][4 * s + 2], ppk[kb][4 * s + 3]};
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
                for (int db = 0; db < DB; db++) {
                    const int chunk = ((db * 4) ^ v_sw) | v_clow;
                    const char* vp = vbuf + (kb * 32 + 16 * s) * ROWB + v_lane_off + chunk * 16;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp + 8 * ROWB));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        }
        }
        __syncthreads();
    }

    // ---- normalise and store: lane (query, half) owns d = 32db + 8g4 + 4hh + 0..3 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (SPLIT) {
        // partials: un-normalised O (fp32), running max and row sum; merged by attn_combine_kernel
        if (qi < p.Sq) {
            const int64_t row = (((int64_t)batch * p.H + head) * p.nsplit + split) * p.Sq + qi;
            if (hh == 0) { p.part_ml[row * 2] = m_run; p.part_ml[row * 2 + 1] = l_tot; }
            float* po = p.part_o + row * D + 4 * hh;
#pragma unroll
            for (int db = 0; db < DB; db++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++)
                    *(f32x4*)(po + 32 * db + 8 * g4) = f32x4{oacc[db][4 * g4], oacc[db][4 * g4 + 1], oacc[db][4 * g4 + 2], oacc[db][4 * g4 + 3]};
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (qi < p.Sq) {
        bf16* op = p.O + (int64_t)batch * p.o_bs + (int64_t)qi * p.o_rs + (int64_t)head * p.o_hs + 4 * hh;
#pragma unroll
        for (int db = 0; db < DB; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = f2bf(oacc[db][4 * g4 + e] * inv);
                *(bf16x4*)(op + 32 * db + 8 * g4) = o;
            }
    }
}

// ---- ViT attention (d = 64, no mask, no score divisor), software-pipelined over 32-key halves ------------------------
// The generic kernel above runs K.Q^T -> softmax -> P.V of a tile back to back in every wave.  At d = 64 the softmax is
// the long pole: 32 scores per lane and 64-key tile against 16 MFMAs, and the vector pipe, not the matrix pipe, sets the
// time (measured per SIMD with scripts/ubench/valu_issue.hip at two waves per SIMD: plain fp32 op 2.4-2.7 cycles,
// v_cvt_pk_bf16_f32 / v_max3 3.5, v_exp_f32 6.5; round-1 counters: ~7.2 vector instructions per score).  So this kernel
// (a) keeps every wave's vector stream free of waits: a half-iteration issues the softmax of half h next to the MFMAs of
//     OTHER halves,  S_{h+1} = K_{h+1} . Q^T (4 MFMAs)  and  O += V_{h-1}^T . P_{h-1} (4 MFMAs); S and P live in two named
//     register sets (even / odd half);
// (b) spends fewer vector instructions per score:
//     * bf16 rounding of a score straight to an fp32 register with ONE v_cvt_pk_bf16_f32 (0, s): the low half of the
//       result is bf16(0) = 0x0000, so the dword IS round_bf16(s) as fp32 -- no unpacking shifts/masks;
//     * no running maximum in the steady state: the reference point m of exp(s - m) is the row maximum of the FIRST
//       half (always <= the true maximum), P = 2^((s - m) log2 e) is rounded to bf16 (which has fp32's exponent range)
//       and accumulated in fp32, so nothing is lost while P stays finite; a half whose row sum reaches 2^60 (any lane:
//       wave-uniform vote) is redone on the spot with the exact maximum and the usual O / l rescale.  Same value of
//       softmax(s) . V up to the rounding of P; the reference's rounding points (modeling_intern_vit.py:225-229: q*scale
//       in bf16, scores in bf16, fp32 softmax, bf16 probabilities into .V) are kept;
// (c) stages K/V through registers (global_load_dwordx4 -> ds_write_b128 one tile later, T14 of the guide): an LDS-DMA
//     piece cost this loop ~140 cycles of issue (in-kernel stamps: 564 cycles per tile for 4 pieces, 20 % of the tile).
// K tiles ride a 3-slot LDS ring two tiles ahead, V tiles a 3-slot ring one tile ahead (a tile's two half-iterations
// read K(t), K(t+1), V(t-1), V(t)); one workgroup barrier per 64-key tile.
struct VitLane {
    int koff[4];       // byte offset of K fragment ks inside a 32-key half (row part included)
    int voff[2];       // byte offset of the transposed V read of d-block db (row part included)
    int hh;
};

// FAST: steady-state form (reference point kept, overflow vote); otherwise the exact-maximum form (first half, edges)
template <bool DO_QK, bool DO_PV, bool MASK, bool FAST>
__device__ __forceinline__ void vit_half(f32x16& s_cur, f32x16& s_next, unsigned (&p_cur)[8], const unsigned (&p_prev)[8],
                                         f32x16 (&oacc)[2], float& m_run, float& l_run, const bf16x8 (&qf)[4],
                                         const char* k_next, const char* v_prev, const VitLane& ln, int key0, int Sk) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    // The vector stream is cut into eight pieces and one MFMA goes in front of each; sched_barrier(0) pins the pieces
    // (left to itself the compiler clumps the MFMAs).
#define VIT_PIN() __builtin_amdgcn_sched_barrier(0)
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix stream operands from LDS
    bf16x8 kf[4];
    bf16x8 vf[2][2];
    if (DO_QK) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) kf[ks] = *(const bf16x8*)(k_next + ln.koff[ks]);
    }
    if (DO_PV) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int db = 0; db < 2; db++) {
                const char* vp = v_prev + s * (16 * 128) + ln.voff[db];
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp + 8 * 128));
                vf[s][db] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
    }
    f32x16 acc;
    if (DO_QK) {
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
    }
    bf16x8 pf[2];
    if (DO_PV) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const u32x4 pw = {p_prev[4 * s], p_prev[4 * s + 1], p_prev[4 * s + 2], p_prev[4 * s + 3]};
            pf[s] = __builtin_bit_cast(bf16x8, pw);
        }
    }
    auto mm = [&](int i) {                 // the i-th of this half-iteration's MFMAs: K.Q^T and V^T.P alternate
        const int j = i >> 1;
        if ((i & 1) == 0) { if (DO_QK) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[j], acc, 0, 0, 0); }
        else { if (DO_PV) oacc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j >> 1][j & 1], pf[j >> 1], oacc[j & 1], 0, 0, 0); }
    };
    // scores of this half, query = lane, keys key0 + (e&3) + 8*(e>>2) + 4*hh
    auto score = [&](int e) -> float {
        float x = s_cur[e];
        if (MASK) { const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * ln.hh; x = key < Sk ? x : -INFINITY; }
        return x;
    };
    float m2, alpha = 1.0f;
    bool rescale = false;
    if (!FAST) {
        // ---- exact form: row maximum on the raw accumulators (rounding is monotonic), across the two half-waves with
        //      one v_permlane32_swap, deferred-rescale vote (threshold 8)
        VIT_PIN();
        mm(0);
        float mraw = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; e += 2) mraw = fmaxf(fmaxf(mraw, score(e)), score(e + 1));
        VIT_PIN();
        mm(1);
        float mloc = rbf1(mraw);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
            mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        rescale = !__all(mloc - m_run <= 8.0f);
        const float m_new = rescale ? fmaxf(m_run, mloc) : m_run;
        alpha = rescale ? __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E) : 1.0f;
        m_run = m_new;
        VIT_PIN();
    }
    m2 = m_run * LOG2E;
    // ---- exponentials of the bf16-rounded scores, row sum, P packed to bf16 (the P.V B operand)
    float ps0 = 0.f, ps1 = 0.f;
    auto expo = [&](int i) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(rbf1(score(2 * i)), LOG2E, -m2));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(rbf1(score(2 * i + 1)), LOG2E, -m2));
        ps0 += p0; ps1 += p1;
        p_cur[i] = pack_bf16(p0, p1);
    };
    if (FAST) {
        // the LDS operands were requested a moment ago: one piece of vector work goes first, then an MFMA per piece
        VIT_PIN();
        expo(0); VIT_PIN();
        mm(0); expo(1); VIT_PIN();
        mm(1); expo(2); VIT_PIN();
        mm(2); expo(3); VIT_PIN();
        mm(3); expo(4); VIT_PIN();
        mm(4); expo(5); VIT_PIN();
        mm(5); expo(6); VIT_PIN();
        mm(6); expo(7); VIT_PIN();
        mm(7);
    } else {
        mm(2); expo(0); VIT_PIN();
        mm(3); expo(1); VIT_PIN();
        mm(4); expo(2); expo(3); VIT_PIN();
        mm(5); expo(4); VIT_PIN();
        mm(6); expo(5); expo(6); VIT_PIN();
        mm(7); expo(7);
    }
    float psum = ps0 + ps1;
    if (DO_QK) s_next = acc;
    __builtin_amdgcn_sched_barrier(0);
    if (FAST) {
        // a row sum of 2^60 or more (a score ~42 above the reference point), or an overflow to +inf, in ANY lane: redo
        // this half with the exact maximum.  The flag is tied to the END of the vector stream (left free, the compiler
        // hoists the branch above the exponentials and splits the half-iteration).
        int bad = __all(psum < 1.152921504606846976e18f) ? 0 : 1;
        asm volatile("" : "+v"(bad), "+v"(psum));
        if (__builtin_amdgcn_readfirstlane(bad)) {
            float mraw = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; e++) mraw = fmaxf(mraw, score(e));
            float mloc = rbf1(mraw);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            m_run = m_new;
            m2 = m_new * LOG2E;
            ps0 = 0.f; ps1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) expo(i);
            psum = ps0 + ps1;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; db++)          // P_{h-1} (already multiplied into O above) was at the old reference point
#pragma unroll
                for (int e = 0; e < 16; e++) oacc[db][e] *= alpha;
        }
        l_run += psum;
    } else {
        l_run = l_run * alpha + psum;
        int flag = rescale ? 1 : 0;
        asm volatile("" : "+v"(flag), "+v"(l_run));
        if (__builtin_amdgcn_readfirstlane(flag)) {
#pragma unroll
            for (int db = 0; db < 2; db++)
#pragma unroll
                for (int e = 0; e < 16; e++) oacc[db][e] *= alpha;
        }
    }
#undef VIT_PIN
}

#ifndef CR_ATTN_VIT_WAVES
#define CR_ATTN_VIT_WAVES 2
#endif
__global__ __launch_bounds__(256, CR_ATTN_VIT_WAVES) void vit_attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 64, ROWB = 128, TILE = 64 * ROWB;      // 8 KiB per K or V tile
    constexpr int VBASE = 3 * TILE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    int qb, head, batch;
    {   // XCD-aware order, as in flash_attn_kernel: the query blocks of one (batch, head) read K/V through one L2
        const int gx = gridDim.x, gy = gridDim.y;
        const int total = gx * gy * (int)gridDim.z;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * (int)blockIdx.z);
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        qb = pid % gx;
        head = (pid / gx) % gy;
        batch = pid / (gx * gy);
    }
    const int Sk = p.Sk;
    const int qi = qb * 128 + wave * 32 + l31;
    const int qi_c = min(qi, p.Sq - 1);
    const bool active = qb * 128 + wave * 32 < p.Sq;          // wave-uniform: a wave of padding rows only stages
#ifdef CR_ATTN_STAMPS
    unsigned long long ph[6];
#define PSTAMP(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define PSTAMP(i)
#endif
    PSTAMP(0);

    const bf16* Kb = p.K + (int64_t)batch * p.k_bs + (int64_t)head * p.k_hs;
    const bf16* Vb = p.V + (int64_t)batch * p.v_bs + (int64_t)head * p.v_hs;
    const int nt = (Sk + 63) / 64;
    const int NH = (Sk + 31) / 32;                            // 32-key halves that hold at least one key
    const bool ragged = (Sk & 31) != 0;

    // staging: a wave owns rows 16*wave .. 16*wave+15 of every tile, as two pieces of 8 rows x 128 B; lane -> (row
    // lane>>3, 16-B chunk lane&7).  The chunk swizzles live on the SOURCE address, the LDS image is lane-linear.
    const int srow0 = wave * 16 + (lane >> 3);
    const int scp = lane & 7;
    // per-lane source pointers of the tile being requested; a full tile later = one 64-bit add of a wave-uniform stride,
    // only a ragged last tile (and requests past the end, which re-load the last tile and are never read) clamp rows
    const bf16* kptr[2];
    const bf16* vptr[2];
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
        const int r = srow0 + ii * 8;
        kptr[ii] = Kb + (int64_t)r * p.k_rs + ((scp ^ kswz<D>(r)) * 8);
        vptr[ii] = Vb + (int64_t)r * p.v_rs + ((scp ^ vswz<D>(r)) * 8);
    }
    const int64_t kstep = 64 * p.k_rs, vstep = 64 * p.v_rs;
    auto load_tile = [&](const bf16* (&ptr)[2], int64_t rs, int64_t step, int kt, bf16x8 (&r)[2]) {
        if (kt * 64 + 64 <= Sk) {                             // wave-uniform
#pragma unroll
            for (int ii = 0; ii < 2; ii++) r[ii] = *(const bf16x8*)(ptr[ii] + (int64_t)kt * step);
        } else {
            const int kc = min(kt, nt - 1);
#pragma unroll
            for (int ii = 0; ii < 2; ii++) {
                const int row = srow0 + ii * 8;
                const int back = max(kc * 64 + row - (Sk - 1), 0);        // rows past the last key read the last key
                r[ii] = *(const bf16x8*)(ptr[ii] + (int64_t)kc * step - (int64_t)back * rs);
            }
        }
    };
    char* my_piece = smem + wave * 2048 + lane * 16;
    auto write_tile = [&](int slot_off, const bf16x8 (&r)[2]) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) *(bf16x8*)(my_piece + slot_off + ii * 1024) = r[ii];
    };

    // ---- prologue: K(0), V(0), K(1) into the rings; K(2) and V(1) on their way in the staging registers; Q fragment
    //      (B operand of K.Q^T: lane (query l31, half hh) holds Q[q][16ks + 8hh .. +7])
    bf16x8 kst[2], vst[2];
    {
        bf16x8 k0[2], v0[2], k1[2];
        load_tile(kptr, p.k_rs, kstep, 0, k0);
        load_tile(vptr, p.v_rs, vstep, 0, v0);
        load_tile(kptr, p.k_rs, kstep, 1, k1);
        write_tile(0, k0);
        write_tile(VBASE, v0);
        write_tile(TILE, k1);
    }
    bf16x8 qf[4];
    {
        const bf16* qp = p.Q + (int64_t)batch * p.q_bs + (int64_t)qi_c * p.q_rs + (int64_t)head * p.q_hs + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qf[ks] = *(const bf16x8*)(qp + ks * 16);
        if (p.q_prescale != 1.0f) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
#pragma unroll
                for (int e = 0; e < 8; e++) qf[ks][e] = f2bf(bf2f(qf[ks][e]) * p.q_prescale);
        }
    }
    load_tile(kptr, p.k_rs, kstep, 2, kst);
    load_tile(vptr, p.v_rs, vstep, 1, vst);
    __syncthreads();

    VitLane ln;
    ln.hh = hh;
    {
        const int k_sw = kswz<D>(l31);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) ln.koff[ks] = l31 * ROWB + (((2 * ks + hh) ^ k_sw) * 16);
        const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
        const int v_lane_row = 4 * hh + tq;
        const int v_sw = vswz<D>(v_lane_row);
        const int v_clow = (g & 1) * 2 + (tp >> 1);
#pragma unroll
        for (int db = 0; db < 2; db++) ln.voff[db] = v_lane_row * ROWB + (tp & 1) * 8 + ((((db * 4) ^ v_sw) | v_clow) * 16);
    }

    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
        for (int e = 0; e < 16; e++) oacc[db][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sA, sB;
    unsigned pA[8], pB[8];
#pragma unroll
    for (int e = 0; e < 16; e++) { sA[e] = 0.f; sB[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; i++) { pA[i] = 0u; pB[i] = 0u; }

    if (active) {                                             // S_0 = K_0 . Q^T
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const bf16x8 kf = *(const bf16x8*)(smem + ln.koff[ks]);
            sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sA, 0, 0, 0);
        }
    }
    int ks_cur = 0, ks_nxt = TILE, ks_nn = 2 * TILE;          // K ring: slots of tiles t, t+1, t+2
    int vs_prev = VBASE + 2 * TILE, vs_cur = VBASE, vs_nxt = VBASE + TILE;   // V ring: tiles t-1, t, t+1
    // tile t: barrier (every wave is done with K(t-1), V(t-2); last tile's ring writes are visible) -> even half ->
    // ring writes + next requests -> odd half.  The staging registers (K(t+2), V(t+1), requested one tile ago) go into the
    // slots the barrier freed, then K(t+3), V(t+2) are requested.  Writing between the halves instead of right behind the
    // barrier spreads the waves' ds_write bursts (stamps: 262 cycles for 4 ds_write_b128 when all 8 waves of the CU write
    // at once, LDS stores run at ~80 B/clk/CU) and is just as safe: neither half of tile t reads the two slots written.
    auto tile_stage = [&](int t) {
        write_tile(ks_nn, kst);
        write_tile(vs_nxt, vst);
        load_tile(kptr, p.k_rs, kstep, t + 3, kst);
        load_tile(vptr, p.v_rs, vstep, t + 2, vst);
    };
    auto rotate = [&]() {
        { const int x = ks_cur; ks_cur = ks_nxt; ks_nxt = ks_nn; ks_nn = x; }
        { const int x = vs_prev; vs_prev = vs_cur; vs_cur = vs_nxt; vs_nxt = x; }
    };
    // `edge` tiles (the first, and those holding one of the last two halves) pick the variant of each half at run time,
    // the tiles in between run the steady-state pair with no decisions
    auto edge_even = [&](int t) {           // softmax of S_A; S_B = K_{2t+1} . Q^T; O += V_{2t-1}^T . P_B
        const int h0 = 2 * t;
        const bool last_e = h0 == NH - 1;
        const char* k_e = smem + ks_cur + 32 * ROWB;          // half 2t+1: rows 32..63 of K(t)
        const char* v_e = smem + vs_prev + 32 * ROWB;         // half 2t-1: rows 32..63 of V(t-1)
        if (!last_e) {
            if (t == 0) vit_half<true, false, false, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
            else vit_half<true, true, false, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
        } else if (ragged) {
            if (t == 0) vit_half<false, false, true, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
            else vit_half<false, true, true, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
        } else {
            if (t == 0) vit_half<false, false, false, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
            else vit_half<false, true, false, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
        }
    };
    auto edge_odd = [&](int t) {            // softmax of S_B; S_A = K_{2t+2} . Q^T (first rows of K(t+1)); O += V_{2t}^T . P_A
        const int h1 = 2 * t + 1;
        if (h1 > NH - 1) return;
        const bool last_o = h1 == NH - 1;
        const char* k_o = smem + ks_nxt;
        const char* v_o = smem + vs_cur;
        if (!last_o) vit_half<true, true, false, false>(sB, sA, pB, pA, oacc, m_run, l_run, qf, k_o, v_o, ln, h1 * 32, Sk);
        else if (ragged) vit_half<false, true, true, false>(sB, sA, pB, pA, oacc, m_run, l_run, qf, k_o, v_o, ln, h1 * 32, Sk);
        else vit_half<false, true, false, false>(sB, sA, pB, pA, oacc, m_run, l_run, qf, k_o, v_o, ln, h1 * 32, Sk);
    };
    const int t_mid_end = (NH - 3) >> 1;                      // last tile whose two halves are both followed by another half
    int t = 0;
    PSTAMP(1);
    if (active) edge_even(0);
    tile_stage(0);
    if (active) edge_odd(0);
    rotate();
    PSTAMP(2);
#ifdef CR_ATTN_STAMPS
    unsigned long long acc_top = 0, acc_even = 0, acc_odd = 0, acc_mid = 0;
#define STAMP(x) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(x)
#endif
    for (t = 1; t <= t_mid_end; t++) {
#ifdef CR_ATTN_STAMPS
        unsigned long long s0, s1, s2, s3, s4;
#endif
        STAMP(s0);
        __syncthreads();
        STAMP(s1);
        if (active) vit_half<true, true, false, true>(sA, sB, pA, pB, oacc, m_run, l_run, qf, smem + ks_cur + 32 * ROWB, smem + vs_prev + 32 * ROWB, ln, 0, Sk);
        STAMP(s2);
        tile_stage(t);
        STAMP(s3);
        if (active) vit_half<true, true, false, true>(sB, sA, pB, pA, oacc, m_run, l_run, qf, smem + ks_nxt, smem + vs_cur, ln, 0, Sk);
        STAMP(s4);
#ifdef CR_ATTN_STAMPS
        acc_top += s1 - s0; acc_even += s2 - s1; acc_mid += s3 - s2; acc_odd += s4 - s3;
#endif
        rotate();
    }
    PSTAMP(3);
    for (; t < nt; t++) {
        __syncthreads();
        if (active) edge_even(t);
        tile_stage(t);
        if (active) edge_odd(t);
        rotate();
    }
    PSTAMP(4);
    // ---- the last half's P.V  (after the rotation above vs_prev holds tile nt-1)
    if (active) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const bool odd_last = ((NH - 1) & 1) != 0;
        const char* vb = smem + vs_prev + (odd_last ? 32 * ROWB : 0);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const u32x4 pw = odd_last ? u32x4{pB[4 * s], pB[4 * s + 1], pB[4 * s + 2], pB[4 * s + 3]}
                                      : u32x4{pA[4 * s], pA[4 * s + 1], pA[4 * s + 2], pA[4 * s + 3]};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int db = 0; db < 2; db++) {
                const char* vp = vb + s * (16 * ROWB) + ln.voff[db];
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)CR_LDS(vp + 8 * ROWB));
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store: lane (query, half) owns d = 32db + 8g4 + 4hh + 0..3
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < p.Sq) {
        bf16* op = p.O + (int64_t)batch * p.o_bs + (int64_t)qi * p.o_rs + (int64_t)head * p.o_hs + 4 * hh;
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = f2bf(oacc[db][4 * g4 + e] * inv);
                *(bf16x4*)(op + 32 * db + 8 * g4) = o;
            }
    }
#ifdef CR_ATTN_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0F70);
    PSTAMP(5);
    if (p.part_o && lane == 0) {
        unsigned long long* d = (unsigned long long*)p.part_o + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + wave) * 12;
        d[0] = acc_top; d[1] = acc_mid; d[2] = acc_even; d[3] = acc_odd;
        for (int i = 0; i < 5; i++) d[4 + i] = ph[i + 1] - ph[i];
        d[9] = active ? 1 : 0;
    }
#endif
}

int launch_vit_attn(const AttnParams& p, hipStream_t stream) {
    constexpr int LDS = 6 * 64 * 128;                        // K ring + V ring, three 8-KiB slots each
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)vit_attn_kernel, LDS)) return CR_ERR_HIP;
    dim3 grid((p.Sq + 127) / 128, p.H, p.B);
#ifdef CR_ATTN_STAMPS
    {   // diagnostic build only: per-wave cycle sums of the steady-state loop, printed as cycles per tile
        static unsigned long long* dbg = nullptr;
        const size_t n = (size_t)grid.x * grid.y * grid.z * 48;
        static size_t cap = 0;
        if (n > cap) { if (dbg) hipFree(dbg); hipMalloc((void**)&dbg, n * 8); cap = n; }
        hipMemsetAsync(dbg, 0, n * 8, stream);
        AttnParams q = p;
        q.part_o = (float*)dbg;
        hipLaunchKernelGGL(vit_attn_kernel, grid, dim3(256), LDS, stream, q);
        hipStreamSynchronize(stream);
        static int calls = 0;
        if (++calls == 3) {
            std::vector<unsigned long long> h(n);
            hipMemcpy(h.data(), dbg, n * 8, hipMemcpyDeviceToHost);
            double a[9] = {0}, ia[5] = {0}; size_t waves = 0, idle = 0;
            for (size_t i = 0; i < n; i += 12) {
                if (h[i + 9]) { for (int k = 0; k < 9; k++) a[k] += (double)h[i + k]; waves++; }
                else { for (int k = 0; k < 5; k++) ia[k] += (double)h[i + 4 + k]; idle++; }
            }
            const int tiles = ((p.Sk + 31) / 32 - 3) / 2;
            fprintf(stderr, "[attn stamps] active waves %zu, steady tiles per wave %d: cycles per tile: barrier %.0f, even half %.0f, ring writes + requests %.0f, odd half %.0f\n",
                    waves, tiles, a[0] / waves / tiles, a[2] / waves / tiles, a[1] / waves / tiles, a[3] / waves / tiles);
            fprintf(stderr, "[attn stamps] cycles per wave: prologue %.0f, tile 0 %.0f, steady loop %.0f, tail tiles %.0f, last P.V + store %.0f;  staging-only waves (%zu): %.0f %.0f %.0f %.0f %.0f\n",
                    a[4] / waves, a[5] / waves, a[6] / waves, a[7] / waves, a[8] / waves, idle, ia[0] / (idle ? idle : 1), ia[1] / (idle ? idle : 1),
                    ia[2] / (idle ? idle : 1), ia[3] / (idle ? idle : 1), ia[4] / (idle ? idle : 1));
        }
        return CR_OK;
    }
#endif
    hipLaunchKernelGGL(vit_attn_kernel, grid, dim3(256), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

template <int D, bool CAUSAL, bool DIV>
int launch_d(const AttnParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * 64 * D * 2;
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)flash_attn_kernel<D, CAUSAL, false, DIV>, LDS)) return CR_ERR_HIP;
    dim3 grid((p.Sq + 127) / 128, p.H, p.B);
    hipLaunchKernelGGL((flash_attn_kernel<D, CAUSAL, false, DIV>), grid, dim3(256), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

template <int D, bool CAUSAL>
int launch_t(const AttnParams& p, hipStream_t stream) {
    return p.s_div != 1.0f ? launch_d<D, CAUSAL, true>(p, stream) : launch_d<D, CAUSAL, false>(p, stream);
}

// out[b][q][h][:] = sum_s exp(m_s - M) O_s / sum_s exp(m_s - M) l_s   (splits in index order: reproducible)
template <int D>
__global__ __launch_bounds__(D) void attn_combine_kernel(const AttnParams p) {
    const int q = blockIdx.x, head = blockIdx.y, batch = blockIdx.z, d = threadIdx.x;
    const int64_t base = ((int64_t)batch * p.H + head) * p.nsplit;
    float M = -INFINITY;
    for (int s = 0; s < p.nsplit; s++) M = fmaxf(M, p.part_ml[((base + s) * p.Sq + q) * 2]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < p.nsplit; s++) {
        const int64_t row = (base + s) * p.Sq + q;
        const float w = __expf(p.part_ml[row * 2] - M);
        L += w * p.part_ml[row * 2 + 1];
        acc += w * p.part_o[row * D + d];
    }
    p.O[(int64_t)batch * p.o_bs + (int64_t)q * p.o_rs + (int64_t)head * p.o_hs + d] = f2bf(acc / L);
}

template <int D, bool DIV>
int launch_split_d(const AttnParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * 64 * D * 2;
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)flash_attn_kernel<D, false, true, DIV>, LDS)) return CR_ERR_HIP;
    hipLaunchKernelGGL((flash_attn_kernel<D, false, true, DIV>), dim3(p.nsplit, p.H, p.B), dim3(256), LDS, stream, p);
    hipLaunchKernelGGL((attn_combine_kernel<D>), dim3(p.Sq, p.H, p.B), dim3(D), 0, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

}  // namespace

size_t attn_split_ws_floats(int B, int H, int Sq, int nsplit, int head_dim) {
    return (size_t)B * H * nsplit * Sq * (head_dim + 2);
}

int launch_flash_attn_split(const AttnParams& p, int head_dim, hipStream_t stream) {
    if (p.Sq <= 0 || p.Sq > 32 || p.H <= 0 || p.B <= 0 || p.nsplit <= 0 || !p.part_ml || !p.part_o) return CR_ERR_ARG;
    if ((p.q_rs & 7) || (p.k_rs & 7) || (p.v_rs & 7)) return CR_ERR_ARG;
    const bool dv = p.s_div != 1.0f;
    if (head_dim == 64) return dv ? launch_split_d<64, true>(p, stream) : launch_split_d<64, false>(p, stream);
    if (head_dim == 128) return dv ? launch_split_d<128, true>(p, stream) : launch_split_d<128, false>(p, stream);
    return CR_ERR_ARG;
}

int launch_flash_attn(const AttnParams& p, int head_dim, bool causal, hipStream_t stream) {
    if (p.Sq <= 0 || (p.Sk <= 0 && !p.sk_arr) || p.H <= 0 || p.B <= 0 || p.kv_group <= 0) return CR_ERR_ARG;
    if ((p.q_rs & 7) || (p.k_rs & 7) || (p.v_rs & 7) || (p.o_rs & 3)) return CR_ERR_ARG;
    if (head_dim == 64 && !causal && p.s_div == 1.0f && p.kv_group == 1 && !p.seq_map && !p.sk_arr) {
        static const bool v1 = [] { const char* e = getenv("CR_ATTN_V1"); return e && atoi(e) != 0; }();   // A/B aid: the unpipelined kernel
        if (!v1) return launch_vit_attn(p, stream);
    }
    if (head_dim == 64) return causal ? launch_t<64, true>(p, stream) : launch_t<64, false>(p, stream);
    if (head_dim == 128) return causal ? launch_t<128, true>(p, stream) : launch_t<128, false>(p, stream);
    return CR_ERR_ARG;
}
