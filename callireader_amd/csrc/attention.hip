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
// Round 2 (commit 3f9c143 holds the code): an in-wave software pipeline over 32-key halves (softmax of half h issued next to
// K.Q^T of half h+1 and P.V of half h-1, two named S / P register sets, MFMAs pinned one per eighth of the vector stream,
// 3-slot K and V rings, one barrier per tile; with LDS-DMA from inline asm, then with register staging) ran the same
// work in 0.543-0.585 ms against 0.52 here: it needs 205 VGPRs = 2 waves per SIMD, and in-kernel stamps put 25 % of a
// wave's 55 k cycles outside the steady loop (prologue 4.9 k, first/last tile 5.8 k, store 3 k) and 350-560 cycles of each
// 2.5 k-cycle tile into ring upkeep (an LDS-DMA piece costs ~140 cycles of issue there, four ds_write_b128 behind a
// barrier ~260).  What did carry over is the lean softmax below: per score one v_cvt_pk_bf16_f32 (0, s), one FMA, one
// v_exp_f32, half a packed add and half a v_cvt_pk (4.0 vector instructions instead of 7.2) -- worth 5-6 % here (0.495 ms),
// because at three waves per SIMD this loop waits on its per-tile barrier and on MFMA results, not on vector issue
// (scripts/ubench/valu_issue.hip: plain fp32 op 2.0-2.4 cycles per SIMD at 3 waves, v_cvt_pk / v_max3 3.0, v_exp 5.7).
// Also measured and dropped: a 3-slot K/V ring (tile kt+2 requested during tile kt, counted vmcnt + raw s_barrier, DMA issued from
// inline asm so that the compiler does not order the ds_reads behind it): 0.517 ms against 0.499-0.510 for this two-slot form in
// the same run -- the fill's latency is not what the loop waits for; workgroups of 2 waves (64 queries; twice the K/V fills per query,
// 10 waves per CU: 0.62 ms) and of 8 waves (256 queries; one workgroup per CU: 0.56 ms) against 0.50-0.51 for these 4.  The lean tile is 138 vector instructions + 16 MFMAs +
// 24 LDS reads per wave; per SIMD the three waves' vector (~1 340 cycles) and matrix (1 536) work would fit 2 880 cycles even
// with no overlap at all, the measured round is ~5 400: the SIMD idles while all three waves sit at their workgroups' barriers.
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
    const int gx = gridDim.x;
    {
        const int gy = gridDim.y;
        const int total = gx * gy * (int)gridDim.z;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * (int)blockIdx.z);
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        bx = pid % gx;
        head = (pid / gx) % gy;
        batch = pid / (gx * gy);
    }
    // causal: the last query blocks have the most keys -- they go first, the short ones fill the tail of the launch
    const int qb = SPLIT ? 0 : (CAUSAL ? gx - 1 - bx : bx);
    const int split = SPLIT ? bx : 0;
    const int kvh = head / p.kv_group;
    int slot = p.seq_map ? p.seq_map[batch] : batch;
    int Sk = p.sk_arr ? p.sk_arr[slot] + p.sk_add : p.Sk;
    int Sq = p.Sq, q_pos0 = p.q_pos0;
    int64_t q_base = (int64_t)batch * p.q_bs, o_base = (int64_t)batch * p.o_bs;
    if (CAUSAL && p.seg) {                                    // several pages' prompts in one launch (wave-uniform)
        const int32_t* sg = p.seg + 4 * batch;
        q_base = (int64_t)sg[0] * p.q_rs; o_base = (int64_t)sg[0] * p.o_rs;
        Sq = sg[1]; q_pos0 = sg[2]; slot = sg[3];
        Sk = q_pos0 + Sq;
        if (qb * 128 >= Sq) return;                           // a shorter page has fewer query blocks
    }
    const int qi = qb * 128 + wave * 32 + l31;
    const int qi_c = min(qi, Sq - 1);

    // ---- Q fragment (B operand of K.Q^T): lane (query l31, half hh) holds Q[q][16ks + 8hh .. +7]
    bf16x8 qf[KS];
    {
        const bf16* qp = p.Q + q_base + (int64_t)qi_c * p.q_rs + (int64_t)head * p.q_hs + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[ks] = *(const bf16x8*)(qp + ks * 16);      // all loads first: with the (run-time) prescale inside this loop hipcc
        if (p.q_prescale != 1.0f) {                                                        // waited for every load in turn (KS dependent round trips per workgroup)
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int e = 0; e < 8; e++) qf[ks][e] = f2bf(bf2f(qf[ks][e]) * p.q_prescale);
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
        const int kmax = q_pos0 + min(qb * 128 + 127, Sq - 1);
        nt = min(nt, kmax / 64 + 1);
    }
    const int qpos = q_pos0 + qi_c;
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
        if (qb * 128 + wave * 32 < Sq) {
        // (a ragged last key tile runs both 32-key halves: the empty one is masked to -inf anyway, and skipping it
        //  cost every tile 16 accumulator-zeroing moves plus two branches on an issue-bound loop)

        // ---- S^T = K . Q^T for two 32-key blocks ----
        f32x16 sacc[2];
        if (!SPLIT) __builtin_amdgcn_s_setprio(0);            // (attention_vit.hip: the softmax / P.V part outranks the K.Q^T MFMAs)
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
        if (!SPLIT) __builtin_amdgcn_s_setprio(1);
        const bool need_mask = (kt * 64 + 64 > Sk) || (CAUSAL && kt * 64 + 63 > q_pos0 + qb * 128 + wave * 32);
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
        // exponentials: plain fp32 FMAs and adds (packed v_pk_fma_f32 / v_pk_add_f32 issue slower beside MFMAs than the two they replace)
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(lo_bf16(spk[kb][i]), LOG2E, -m2));
                const float p1 = __builtin_amdgcn_exp2f(fmaf(hi_bf16(spk[kb][i]), LOG2E, -m2));
                ps0 += p0; ps1 += p1;
                ppk[kb][i] = pack_bf16(p0, p1);
            }
        const float psum = ps0 + ps1;
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
        if (qi < Sq) {
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
    if (qi < Sq) {
        bf16* op = p.O + o_base + (int64_t)qi * p.o_rs + (int64_t)head * p.o_hs + 4 * hh;
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
// A wave's lane s fetches split s's (m, l) -- one round trip for up to 64 splits, maximum and weights by shuffles -- and the O rows are
// loaded eight at a time with independent loads (a plain loop over the run-time split count waited for every load in turn: 13
// dependent round trips per launch at 3 300 cached tokens, 11.5 us x 32 layers per decode step).
template <int D>
__global__ __launch_bounds__(D) void attn_combine_kernel(const AttnParams p) {
    const int q = blockIdx.x, head = blockIdx.y, batch = blockIdx.z, d = threadIdx.x, lane = d & 63;
    const int64_t base = ((int64_t)batch * p.H + head) * p.nsplit;
    float M = -INFINITY;
    for (int s0 = 0; s0 < p.nsplit; s0 += 64) {
        const int s = s0 + lane;
        M = fmaxf(M, wave_max(s < p.nsplit ? p.part_ml[((base + s) * p.Sq + q) * 2] : -INFINITY));
    }
    float L = 0.f, acc = 0.f;
    for (int s0 = 0; s0 < p.nsplit; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < p.nsplit;
        const int64_t row = (base + (in ? s : 0)) * p.Sq + q;
        const float w = in ? __expf(p.part_ml[row * 2] - M) : 0.f;
        const float wl = in ? w * p.part_ml[row * 2 + 1] : 0.f;
        const int n = min(64, p.nsplit - s0);
        for (int j0 = 0; j0 < n; j0 += 8) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = j0 + j < n ? p.part_o[((base + s0 + j0 + j) * p.Sq + q) * D + d] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {                    // index order; lanes past the count carry weight 0
                L += __shfl(wl, j0 + j, 64);
                acc += __shfl(w, j0 + j, 64) * o[j];
            }
        }
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
    if (decode_attn_supported(p, head_dim)) {                 // attention_decode.hip writes the same partials; the combine is this file's
        if (launch_decode_attn(p, stream) != CR_OK) return CR_ERR_HIP;
        hipLaunchKernelGGL((attn_combine_kernel<128>), dim3(p.Sq, p.H, p.B), dim3(128), 0, stream, p);
        return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
    }
    if (head_dim == 64) return dv ? launch_split_d<64, true>(p, stream) : launch_split_d<64, false>(p, stream);
    if (head_dim == 128) return dv ? launch_split_d<128, true>(p, stream) : launch_split_d<128, false>(p, stream);
    return CR_ERR_ARG;
}

int launch_flash_attn(const AttnParams& p, int head_dim, bool causal, hipStream_t stream) {
    if (p.Sq <= 0 || (p.Sk <= 0 && !p.sk_arr && !p.seg) || p.H <= 0 || p.B <= 0 || p.kv_group <= 0 || (p.seg && !causal)) return CR_ERR_ARG;
    if ((p.q_rs & 7) || (p.k_rs & 7) || (p.v_rs & 7) || (p.o_rs & 3)) return CR_ERR_ARG;
    if (vit_attn_supported(p, head_dim, causal)) return launch_vit_attn(p, stream);
    if (head_dim == 64) return causal ? launch_t<64, true>(p, stream) : launch_t<64, false>(p, stream);
    if (head_dim == 128) return causal ? launch_t<128, true>(p, stream) : launch_t<128, false>(p, stream);
    return CR_ERR_ARG;
}
