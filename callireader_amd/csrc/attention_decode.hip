// Decode attention for gfx950: one new token per sequence over its KV cache (InternLM2 GQA: the 4 query heads of a KV group x 128, modeling_internlm2.py:393-410),
// as a STREAMING kernel on the vector pipe -- the split half of launch_flash_attn_split; the partials it writes are attn_combine_kernel's, unchanged.
//
// Why not the matrix-core kernel (attention.hip: flash_attn_kernel<128, false, true>): decode attention is a byte stream (131 KB of K / V per cached token and
// layer, 4 flops per byte).  That kernel stages 64-key tiles through two LDS buffers of 32 KB, so a CU holds two workgroups with ONE tile each in flight, and a
// split walks its four tiles behind four barriers: at 8 rows x 3 200 keys 832 workgroups move 106 MB in 25.5 us (4.2 TB/s; one row 10.3 us), and every variant
// of that structure measured the same or worse (profiles/round5/04_*: splits of two tiles requested at once, the four waves dividing the keys with a merge in
// LDS).  What bounds it is bytes in flight per CU, and LDS is what caps them.
//
// Here nothing is staged: a workgroup = one split of 256 keys of one (row, KV head), a wave = 64 of them, and a lane owns 16 bytes of every fourth key --
// lane = (key sub-index 0..3, 16-byte chunk 0..15), so one load instruction reads 4 consecutive keys = ONE contiguous KiB -- with all 16 K and 16 V loads of the
// wave issued up front (512 bytes per lane, 32 KB per wave, 256 KB per CU at two workgroups per CU: eight times the old form).  Per key: q . k for the four
// query heads by v_dot2c_f32_bf16 on the lane's 8 dimensions and four DPP rotate-adds over the 16 lanes of the key; the scores' rounding points are the
// reference's (bf16, / sqrt(128), bf16); each 16-lane group keeps ITS 16 keys' scores, so its softmax needs no running maximum (one exponential per lane and
// key: lane c computes query head c & 3, the quad shares them by DPP); P.V by fp32 FMAs on bf16-rounded probabilities.  The 16 (wave, group) partials of the
// workgroup meet in LDS and are merged in index order: a row's result depends on its own key count only, never on the batch.
#include <stdint.h>
#include <stdlib.h>

#include "attention.hpp"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int HD = 128, NQ = 4, IT = 16;      // head dimension, query heads per KV head, keys per lane group and wave (4 groups x 16 = 64 keys per wave)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float rbf1(float x) {           // round to bf16, as an fp32 value, in one v_cvt_pk_bf16_f32 (attention.hip)
    const f32x2_t v = {0.f, x};
    return __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)));
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, left in every lane (rotations: the order of the adds is the same in every lane group of every launch)
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp<0x128>(v);      // row_ror:8
    v += dpp<0x124>(v);      // row_ror:4
    v += dpp<0x122>(v);      // row_ror:2
    v += dpp<0x121>(v);      // row_ror:1
    return v;
}

constexpr int FOLD_MAX_SPLITS = 8;       // K-slices of the wqkv partial-sum GEMM the folded prologue takes (gemm_skinny.hip: partial_geom gives 2..8)

// FOLD: RoPE + split + the cache write of the new token inside this kernel (round-5 verdict, item 3: the 9..64-row decode step ran rope_split_kernel as a
// launch of its own, 6.9 us of a 13-row layer's 133).  Threads 0..95 = (slot 0..5 [4 q heads | k | v], 16-byte chunk c) of this (row, KV group) -- exactly
// rope_split_kernel's thread -- request their chunk of every K-slice BEFORE the K / V stream (loads return in order: the sums then wait for 16 loads, not for the
// stream), add the slices in slice order from 0.f, round once, and rotate with the partner chunk (c + 8) & 15 fetched by a DPP row rotation (the 16 chunks of a
// slot are one DPP row); the six bf16 rows meet in 1.5 KiB of LDS behind one barrier.  q is read from there; the workgroup whose split holds the new position also
// stores the K / V row to the cache and puts it into the registers of the lanes that would have loaded it (every lane at or past the new position: the cache row
// itself is not written yet when they load it).  The same fp32 sums, the same roundings: a row's bits do not depend on which form its batch takes.
template <bool DIV, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void decode_attn_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float s_o[16][NQ][HD];      // the workgroup's 16 partials: un-normalised O ...
    __shared__ float s_ml[16][NQ][2];                                   // ... maximum and row sum
    __shared__ __attribute__((aligned(16))) bf16x8 s_qkv[FOLD ? 6 : 1][16];   // FOLD: the rotated q heads, the new K row, the new V row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, head = blockIdx.y, batch = blockIdx.z;
    const int slot = p.seq_map ? p.seq_map[batch] : batch;
    const int Sk = p.sk_arr ? p.sk_arr[slot] + p.sk_add : p.Sk;
    const int64_t prow = (((int64_t)batch * p.H + head) * p.nsplit + split) * NQ;
    if (split * 256 >= Sk) {                                  // a shorter row's empty split: (-inf, 0, 0), as the matrix-core kernel leaves it
        for (int idx = tid; idx < NQ * HD; idx += 256) p.part_o[prow * HD + idx] = 0.f;
        if (tid < NQ) { p.part_ml[(prow + tid) * 2] = -INFINITY; p.part_ml[(prow + tid) * 2 + 1] = 0.f; }
        return;
    }
    const int ksub = lane >> 4, c = lane & 15, own = c & 3;
    const int key0 = split * 256 + wave * 64 + ksub;          // this lane's keys: key0 + 4 i
    // one uniform 64-bit base per (slot, head) plane + a 32-bit per-lane element offset (a plane is max_tokens x 128 elements: far below 2^31): the 64 x 64-bit
    // products of `key * stride` per load were a fifth of the kernel's vector instructions
    const bf16* Kb = p.K + (int64_t)slot * p.k_bs + (int64_t)head * p.k_hs;
    const bf16* Vb = p.V + (int64_t)slot * p.v_bs + (int64_t)head * p.v_hs;
    const unsigned k_rs = (unsigned)p.k_rs, v_rs = (unsigned)p.v_rs, lane_off = (unsigned)c * 8u;
    // the query rows first, then all 32 K / V loads behind them, and NOTHING else in between: loads return in order, so the first use of q waits for four
    // loads, not for the stream (left to itself the scheduler consumed q before it had issued the stream: a round trip with nothing in flight)
    bf16x8 q[NQ];
    f32x4 pv[FOLD ? FOLD_MAX_SPLITS : 1][2];
    bf16x8 rc = {}, rs = {};
    const int fslot = tid >> 4;                               // FOLD: threads 0..95 = (slot, chunk c)
    if (!FOLD) {
        const bf16* qp = p.Q + (int64_t)batch * p.q_bs + (int64_t)head * p.q_hs + lane_off;
#pragma unroll
        for (int h = 0; h < NQ; h++) q[h] = *(const bf16x8*)(qp + (int64_t)h * p.q_rs);
    } else if (fslot < 6) {
        const float* pp = p.qkv_part + (int64_t)batch * p.qkv_ld + (head * 6 + fslot) * HD + c * 8;
        const int64_t sstep = (int64_t)gridDim.z * p.qkv_ld;
#pragma unroll
        for (int s = 0; s < FOLD_MAX_SPLITS; s++) {
            const f32x4* ps = (const f32x4*)(pp + (int64_t)min(s, p.qkv_splits - 1) * sstep);
            pv[s][0] = ps[0]; pv[s][1] = ps[1];
        }
        rc = *(const bf16x8*)(p.rope_cos + (int64_t)(Sk - 1) * HD + c * 8);
        rs = *(const bf16x8*)(p.rope_sin + (int64_t)(Sk - 1) * HD + c * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kr[IT], vr[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) kr[i] = __builtin_nontemporal_load((const bf16x8*)(Kb + ((unsigned)min(key0 + 4 * i, Sk - 1) * k_rs + lane_off)));
#pragma unroll
    for (int i = 0; i < IT; i++) vr[i] = __builtin_nontemporal_load((const bf16x8*)(Vb + ((unsigned)min(key0 + 4 * i, Sk - 1) * v_rs + lane_off)));
    __builtin_amdgcn_sched_barrier(0);
    if (FOLD) {
        const bool new_here = (Sk - 1) >> 8 == split;         // this workgroup's split holds the new token's position (workgroup-uniform)
        if (fslot < 6) {
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < FOLD_MAX_SPLITS; s++)
                if (s < p.qkv_splits) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { a[e] += pv[s][0][e]; a[4 + e] += pv[s][1][e]; }
                }
            bf16x8 x;
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = f2bf(a[e]);
            bf16x8 y = x;
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            const i32x4_t xi = __builtin_bit_cast(i32x4_t, x);
            i32x4_t xpi;
#pragma unroll
            for (int e = 0; e < 4; e++) xpi[e] = __builtin_amdgcn_update_dpp(0, xi[e], 0x128, 0xf, 0xf, true);      // row_ror:8: chunk (c + 8) & 15 of the same slot
            if (fslot < 5) {
                const bf16x8 xp = __builtin_bit_cast(bf16x8, xpi);
                const float sign = c < 8 ? -1.0f : 1.0f;       // rotate_half: (-x2, x1)
#pragma unroll
                for (int e = 0; e < 8; e++) y[e] = f2bf(rbf(bf2f(x[e]) * bf2f(rc[e])) + rbf(sign * bf2f(xp[e]) * bf2f(rs[e])));
            }
            s_qkv[fslot][c] = y;
            if (new_here && fslot >= 4) {                       // the cache row of the new token (read by the next steps)
                bf16* dst = const_cast<bf16*>(fslot == 4 ? Kb : Vb) + ((unsigned)(Sk - 1) * (fslot == 4 ? k_rs : v_rs) + lane_off);
                *(bf16x8*)dst = y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < NQ; h++) q[h] = s_qkv[h][c];
        if (new_here) {
            const bf16x8 kn = s_qkv[4][c], vn = s_qkv[5][c];
#pragma unroll
            for (int i = 0; i < IT; i++)
                if (key0 + 4 * i >= Sk - 1) { kr[i] = kn; vr[i] = vn; }
        }
    }
    const float inv_div = 1.0f / p.s_div;

    // ---- scores of this lane's own query head for its group's 16 keys ----
    float sc[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) {
        float d[NQ];
#pragma unroll
        for (int h = 0; h < NQ; h++) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bf16x2_t qq = {q[h][2 * e], q[h][2 * e + 1]}, kk = {kr[i][2 * e], kr[i][2 * e + 1]};
                a = __builtin_amdgcn_fdot2_f32_bf16(qq, kk, a, false);
            }
            d[h] = row_sum16(a);
        }
        float s = own == 0 ? d[0] : own == 1 ? d[1] : own == 2 ? d[2] : d[3];
        s = rbf1(s);                                          // scores are bf16 (:393), divided by sqrt(d) -> bf16
        if (DIV) s = rbf1(s * inv_div);
        sc[i] = key0 + 4 * i < Sk ? s : -INFINITY;
    }
    float m = sc[0];
#pragma unroll
    for (int i = 1; i < IT; i++) m = fmaxf(m, sc[i]);
    const float m2 = (m == -INFINITY ? 0.f : m) * LOG2E;      // a group with no key below Sk: every P is exp(-inf) = 0
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < IT; i++) {
        const float pe = __builtin_amdgcn_exp2f(fmaf(sc[i], LOG2E, -m2));
        l += pe;                                              // fp32 softmax (:409): the sum of the un-rounded exponentials
        sc[i] = rbf1(pe);                                     // probabilities go to bf16 before .V
    }

    // ---- O[h][8 dims of this lane] += P[h] . V ----
    float o[NQ][8];
#pragma unroll
    for (int h = 0; h < NQ; h++)
#pragma unroll
        for (int e = 0; e < 8; e++) o[h][e] = 0.f;
#pragma unroll
    for (int i = 0; i < IT; i++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (float)vr[i][e];
        const float p0 = dpp<0x00>(sc[i]), p1 = dpp<0x55>(sc[i]), p2 = dpp<0xaa>(sc[i]), p3 = dpp<0xff>(sc[i]);     // quad_perm: lane (quad, h) holds head h's P
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[0][e] = fmaf(p0, v[e], o[0][e]);
            o[1][e] = fmaf(p1, v[e], o[1][e]);
            o[2][e] = fmaf(p2, v[e], o[2][e]);
            o[3][e] = fmaf(p3, v[e], o[3][e]);
        }
    }

    // ---- the 16 partials of the workgroup (wave, group) in LDS, merged in index order ----
    const int j = wave * 4 + ksub;
#pragma unroll
    for (int h = 0; h < NQ; h++) {
        *(f32x4*)&s_o[j][h][c * 8] = f32x4{o[h][0], o[h][1], o[h][2], o[h][3]};
        *(f32x4*)&s_o[j][h][c * 8 + 4] = f32x4{o[h][4], o[h][5], o[h][6], o[h][7]};
    }
    if (c < NQ) { s_ml[j][c][0] = m; s_ml[j][c][1] = l; }     // lane c of a group holds head c's (m, l) (so do lanes c + 4, c + 8, c + 12)
    __syncthreads();
    {
        const int h = tid >> 6, d0 = (tid & 63) * 2;
        float M = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 16; jj++) M = fmaxf(M, s_ml[jj][h][0]);
        float L = 0.f, a0 = 0.f, a1 = 0.f;
        if (M != -INFINITY) {
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const float mj = s_ml[jj][h][0];
                const float w = mj == -INFINITY ? 0.f : __expf(mj - M);
                L += w * s_ml[jj][h][1];
                a0 += w * s_o[jj][h][d0];
                a1 += w * s_o[jj][h][d0 + 1];
            }
        }
        *(f32x2_t*)(p.part_o + (prow + h) * HD + d0) = f32x2_t{a0, a1};
        if ((tid & 63) == 0) { p.part_ml[(prow + h) * 2] = M; p.part_ml[(prow + h) * 2 + 1] = L; }
    }
}

}  // namespace

// the shape the kernel is written for: d = 128, four query rows per (batch, head) (the GQA group), splits of 256 keys (ATTN_SPLIT_TILES = 4)
bool decode_attn_supported(const AttnParams& p, int head_dim) {
    static const bool off = [] { const char* e = getenv("CR_DECODE_ATTN"); return e && e[0] == '0'; }();      // A/B aid: the matrix-core split kernel
    // kv_group == 1: the kernel indexes K / V by blockIdx.y itself, i.e. H counts KV heads and the GQA group is the Sq = 4 query rows
    return !off && !p.force_matrix_core && p.kv_group == 1 && head_dim == HD && p.Sq == NQ && ATTN_SPLIT_TILES == 4 && p.q_prescale == 1.0f && p.part_ml && p.part_o && p.nsplit > 0 &&
           (p.q_rs & 7) == 0 && (p.q_hs & 7) == 0 && (p.q_bs & 7) == 0 && (p.k_rs & 7) == 0 && (p.v_rs & 7) == 0 && (p.k_hs & 7) == 0 && (p.v_hs & 7) == 0 &&
           (p.k_bs & 7) == 0 && (p.v_bs & 7) == 0 && (p.sk_arr || p.Sk > 0) && p.k_rs > 0 && p.v_rs > 0 && p.k_rs < 65536 && p.v_rs < 65536 &&
           (int64_t)p.nsplit * 256 * (p.k_rs > p.v_rs ? p.k_rs : p.v_rs) < (1ll << 32) &&      // key * row stride stays in the 32-bit per-lane offset
           (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V) & 15) == 0 && ((uintptr_t)p.part_o & 7) == 0;      // 16-byte loads, 8-byte partial stores
}

// the folded form on top: partial sums of the [H groups][4 q | k | v][128] row, at most FOLD_MAX_SPLITS slices, one new token per row at position sk_arr[slot]
bool decode_attn_fold_supported(const AttnParams& p, int head_dim) {
    return decode_attn_supported(p, head_dim) && p.qkv_part && p.rope_cos && p.rope_sin && p.qkv_splits >= 1 && p.qkv_splits <= FOLD_MAX_SPLITS &&
           p.sk_arr && p.sk_add == 1 && p.k_rs == HD && p.v_rs == HD && (p.qkv_ld & 3) == 0 && p.qkv_ld >= (int64_t)p.H * 6 * HD &&
           (((uintptr_t)p.qkv_part | (uintptr_t)p.rope_cos | (uintptr_t)p.rope_sin) & 15) == 0;
}

int launch_decode_attn(const AttnParams& p, hipStream_t stream) {
    const dim3 grid(p.nsplit, p.H, p.B);
    if (p.qkv_part) {
        if (!decode_attn_fold_supported(p, HD)) return CR_ERR_ARG;
        if (p.s_div != 1.0f) hipLaunchKernelGGL((decode_attn_kernel<true, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((decode_attn_kernel<false, true>), grid, dim3(256), 0, stream, p);
    } else if (p.s_div != 1.0f) hipLaunchKernelGGL((decode_attn_kernel<true, false>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((decode_attn_kernel<false, false>), grid, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}
