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
#include <stdlib.h>

#include <type_traits>

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
        unsigned ppk[2][8];                                   // P as packed bf16 pairs: the PV B-operand, 4 dwords per fragment
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
// The kernel above runs K.Q^T -> softmax -> P.V of a tile back to back in every wave: each stage waits for the one
// before it (round-1 counters: per wave and 64-key tile ~945 cycles of vector issue + 512 of MFMA + ~200 of LDS/scalar
// issue ADD UP to the ~1690 observed).  At d = 64 the softmax is the long pole (32 scores per lane and tile at ~6 vector
// issue slots each against 16 MFMAs), so here every wave keeps its vector stream busy and sprinkles the MFMAs of OTHER
// halves into it:
//   half-iteration h:   vector: softmax of half h (scores S_h computed one half-iteration ago)
//                       matrix: S_{h+1} = K_{h+1} . Q^T (4 MFMAs)  and  O += V_{h-1}^T . P_{h-1} (4 MFMAs)
// Nothing a half-iteration issues depends on what it issues itself, so neither pipe waits for the other; the two or
// three waves of a SIMD fill each other's issue gaps.  S and P live in two named register sets (even / odd half).
// K tiles ride a 3-slot LDS ring two tiles ahead, V tiles a 3-slot ring one tile ahead (a half-iteration of tile t reads
// K(t), K(t+1), V(t-1), V(t)); one workgroup barrier per 64-key tile.  Rounding points as above
// (modeling_intern_vit.py:225-229).  The O / l rescale of the online softmax is deferred (threshold 8) and applied at
// the end of the half-iteration that decided it, after the P.V of the previous half (whose P is at the old maximum).
#ifndef CR_ATTN_VIT_SCHED
#define CR_ATTN_VIT_SCHED 1      // 0: program order left to the compiler; 1: MFMAs spaced through the vector stream
#endif

// 1 KiB LDS-DMA piece issued from inline asm: hipcc keeps no count of it, so it neither drains the ring with a
// vmcnt(0) in front of the next LDS read (it cannot tell the slots apart) nor at a barrier; the kernel waits for its own
// DMA once per tile, right before the barrier that publishes the tile.  M0 is saved and restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct VitLane {
    int koff[4];       // byte offset of K fragment ks inside a 32-key half (row part included)
    int voff[2];       // byte offset of the transposed V read of d-block db (row part included)
    int hh;
};

template <bool DO_QK, bool DO_PV, bool MASK>
__device__ __forceinline__ void vit_half(f32x16& s_cur, f32x16& s_next, unsigned (&p_cur)[8], const unsigned (&p_prev)[8],
                                         f32x16 (&oacc)[2], float& m_run, float& l_run, const bf16x8 (&qf)[4],
                                         const char* k_next, const char* v_prev, const VitLane& ln, int key0, int Sk) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    // The vector stream is cut into eight pieces and one MFMA goes in front of each; sched_barrier(0) pins the pieces
    // (CR_ATTN_VIT_SCHED == 0 leaves the order to the compiler, which clumps the MFMAs).
#if CR_ATTN_VIT_SCHED == 1
#define VIT_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define VIT_PIN()
#endif
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix stream operands from LDS (consumed one half-iteration's worth of vector work later)
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
    VIT_PIN();
    // ---- piece 0: reference rounding of the scores (pairs, one v_cvt_pk_bf16_f32 each) and the row maximum on the raw
    //      accumulators (rounding is monotonic); query = lane, keys key0 + (e&3) + 8*(e>>2) + 4*hh
    mm(0);
    float mraw = -INFINITY;
    unsigned spk[8];
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        float s0 = s_cur[e], s1 = s_cur[e + 1];
        if (MASK) {
            const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * ln.hh;
            s0 = key < Sk ? s0 : -INFINITY;
            s1 = key + 1 < Sk ? s1 : -INFINITY;
        }
        mraw = fmaxf(fmaxf(mraw, s0), s1);
        spk[e >> 1] = pack_bf16(s0, s1);
    }
    VIT_PIN();
    // ---- piece 1: maximum across the two half-waves (one v_permlane32_swap), deferred-rescale vote, new reference point
    mm(1);
    float mloc = lo_bf16(pack_bf16(mraw, mraw));
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const bool rescale = !__all(mloc - m_run <= 8.0f);
    const float m_new = rescale ? fmaxf(m_run, mloc) : m_run;
    const float m2 = m_new * LOG2E;
    const float alpha = rescale ? __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E) : 1.0f;
    m_run = m_new;
    VIT_PIN();
    // ---- pieces 2..7: exponentials, row sum, P packed to bf16 (the P.V B operand)
    float ps0 = 0.f, ps1 = 0.f;
    auto expo = [&](int i) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(lo_bf16(spk[i]), LOG2E, -m2));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(hi_bf16(spk[i]), LOG2E, -m2));
        ps0 += p0; ps1 += p1;
        p_cur[i] = pack_bf16(p0, p1);
    };
    mm(2); expo(0); VIT_PIN();
    mm(3); expo(1); VIT_PIN();
    mm(4); expo(2); expo(3); VIT_PIN();
    mm(5); expo(4); VIT_PIN();
    mm(6); expo(5); expo(6); VIT_PIN();
    mm(7); expo(7);
    l_run = l_run * alpha + (ps0 + ps1);
    if (DO_QK) s_next = acc;
    __builtin_amdgcn_sched_barrier(0);
    // the (rare, wave-uniform) rescale is tied to the END of the vector stream: left free, the compiler hoists the branch
    // above the exponentials, which cuts the half-iteration in two and leaves the MFMAs in a clump of their own
    int flag = rescale ? 1 : 0;
    asm volatile("" : "+v"(flag), "+v"(l_run));
    if (__builtin_amdgcn_readfirstlane(flag)) {        // P_{h-1} (already multiplied into O above) was at the old maximum
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int e = 0; e < 16; e++) oacc[db][e] *= alpha;
    }
#undef VIT_PIN
}

__global__ __launch_bounds__(256, 2) void vit_attn_kernel(const AttnParams p) {
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

    const bf16* Kb = p.K + (int64_t)batch * p.k_bs + (int64_t)head * p.k_hs;
    const bf16* Vb = p.V + (int64_t)batch * p.v_bs + (int64_t)head * p.v_hs;

    // staging: a wave fills rows 16*wave .. 16*wave+15 of a tile with two 1-KiB DMA pieces (8 rows x 128 B each);
    // the 16-B chunk swizzles live on the SOURCE address, the LDS image is lane-linear
    const int srow0 = wave * 16 + (lane >> 3);
    const int scp = lane & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)CR_LDS(smem);
    // per-lane source pointers of tile 0 are computed once; a full tile adds a wave-uniform offset, only the ragged last
    // tile re-derives clamped rows
    const bf16* kbase[2];
    const bf16* vbase[2];
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
        const int r = srow0 + ii * 8;
        kbase[ii] = Kb + (int64_t)r * p.k_rs + ((scp ^ kswz<D>(r)) * 8);
        vbase[ii] = Vb + (int64_t)r * p.v_rs + ((scp ^ vswz<D>(r)) * 8);
    }
    auto stage = [&](const bf16* base, int64_t rs, int slot_off, int kt, bool is_v) {
        const bool full = kt * 64 + 64 <= Sk;
        const int64_t toff = (int64_t)kt * 64 * rs;              // scalar
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const bf16* src;
            if (full) src = (is_v ? vbase[ii] : kbase[ii]) + toff;
            else {
                const int r = srow0 + ii * 8;
                const int key = min(kt * 64 + r, Sk - 1);
                const int sw = is_v ? vswz<D>(r) : kswz<D>(r);
                src = base + (int64_t)key * rs + ((scp ^ sw) * 8);
            }
            glds16_asm(src, __builtin_amdgcn_readfirstlane(lds0 + slot_off + (wave * 2 + ii) * 1024));
        }
    };

    const int nt = (Sk + 63) / 64;
    const int NH = (Sk + 31) / 32;                            // 32-key halves that hold at least one key
    const bool ragged = (Sk & 31) != 0;

    // ---- prologue: K(0), V(0), K(1) on their way, then the Q fragment (B operand of K.Q^T: lane (query l31, half hh)
    //      holds Q[q][16ks + 8hh .. +7]), then ONE compiler-visible vmcnt(0): hipcc must know the Q loads have landed,
    //      or it carries "Q may be pending" into the loop and puts counted vmcnt waits in front of the K.Q^T MFMAs --
    //      which on the hardware counter wait for this kernel's own (asm, uncounted) DMA
    stage(Kb, p.k_rs, 0, 0, false);
    stage(Vb, p.v_rs, VBASE, 0, true);
    if (nt > 1) stage(Kb, p.k_rs, TILE, 1, false);
    bf16x8 qf[4];
    {
        const bf16* qp = p.Q + (int64_t)batch * p.q_bs + (int64_t)qi_c * p.q_rs + (int64_t)head * p.q_hs + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
    if (p.q_prescale != 1.0f) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) qf[ks][e] = f2bf(bf2f(qf[ks][e]) * p.q_prescale);
    }
    __builtin_amdgcn_s_barrier();

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
    // one tile = its even half then its odd half; `edge` tiles (the first, and those holding one of the last two halves)
    // pick the variant of each half at run time, the tiles in between run the steady-state pair with no decisions
    auto tile_sync_stage = [&](int t) {
        if (t > 0) {                                          // K(t+1), V(t) landed; every wave is done with K(t-1), V(t-2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (t + 2 < nt) stage(Kb, p.k_rs, ks_nn, t + 2, false);
        if (t + 1 < nt) stage(Vb, p.v_rs, vs_nxt, t + 1, true);
    };
    auto rotate = [&]() {
        { const int x = ks_cur; ks_cur = ks_nxt; ks_nxt = ks_nn; ks_nn = x; }
        { const int x = vs_prev; vs_prev = vs_cur; vs_cur = vs_nxt; vs_nxt = x; }
    };
    auto edge_tile = [&](int t) {
        const int h0 = 2 * t;
        const bool last_e = h0 == NH - 1;
        const char* k_e = smem + ks_cur + 32 * ROWB;          // half 2t+1: rows 32..63 of K(t)
        const char* v_e = smem + vs_prev + 32 * ROWB;         // half 2t-1: rows 32..63 of V(t-1)
        // even half: softmax of S_A; S_B = K_{2t+1} . Q^T; O += V_{2t-1}^T . P_B
        if (!last_e) {
            if (t == 0) vit_half<true, false, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
            else vit_half<true, true, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
        } else if (ragged) {
            if (t == 0) vit_half<false, false, true>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
            else vit_half<false, true, true>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
        } else {
            if (t == 0) vit_half<false, false, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
            else vit_half<false, true, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, k_e, v_e, ln, h0 * 32, Sk);
        }
        if (!last_e) {
            // odd half: softmax of S_B; S_A = K_{2t+2} . Q^T (first rows of K(t+1)); O += V_{2t}^T . P_A
            const bool last_o = h0 + 1 == NH - 1;
            const char* k_o = smem + ks_nxt;
            const char* v_o = smem + vs_cur;
            if (!last_o) vit_half<true, true, false>(sB, sA, pB, pA, oacc, m_run, l_run, qf, k_o, v_o, ln, (h0 + 1) * 32, Sk);
            else if (ragged) vit_half<false, true, true>(sB, sA, pB, pA, oacc, m_run, l_run, qf, k_o, v_o, ln, (h0 + 1) * 32, Sk);
            else vit_half<false, true, false>(sB, sA, pB, pA, oacc, m_run, l_run, qf, k_o, v_o, ln, (h0 + 1) * 32, Sk);
        }
    };
    const int t_mid_end = (NH - 3) >> 1;                      // last tile whose two halves are both followed by another half
    int t = 0;
    tile_sync_stage(0);
    if (active) edge_tile(0);
    rotate();
    for (t = 1; t <= t_mid_end; t++) {
        tile_sync_stage(t);
        if (active) {
            vit_half<true, true, false>(sA, sB, pA, pB, oacc, m_run, l_run, qf, smem + ks_cur + 32 * ROWB, smem + vs_prev + 32 * ROWB, ln, 0, Sk);
            vit_half<true, true, false>(sB, sA, pB, pA, oacc, m_run, l_run, qf, smem + ks_nxt, smem + vs_cur, ln, 0, Sk);
        }
        rotate();
    }
    for (; t < nt; t++) {
        tile_sync_stage(t);
        if (active) edge_tile(t);
        rotate();
    }
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
}

int launch_vit_attn(const AttnParams& p, hipStream_t stream) {
    constexpr int LDS = 6 * 64 * 128;                        // K ring + V ring, three 8-KiB slots each
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)vit_attn_kernel, LDS)) return CR_ERR_HIP;
    dim3 grid((p.Sq + 127) / 128, p.H, p.B);
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
