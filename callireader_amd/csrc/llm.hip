// Language-model stage: InternLM2.5-7B prefill + batched greedy decode over a pre-allocated KV cache.
//   reference: InternVL/modeling_internlm2.py:129-143 (RMSNorm), :233-247 (RoPE), :250-264 (SwiGLU),
//              :341-426 (attention), :621-681 (layer), :854-984 (model), :1022-1110 (LM head),
//              :1112-1149 (prepare_inputs_for_generation); greedy loop = transformers 4.45.2 _sample.
//
// Per layer: RMSNorm -> GEMM wqkv -> RoPE + split (q to a dense buffer, k/v straight into the cache)
//            -> flash attention (causal GQA in prefill; 4 heads of a KV group as the "rows" in decode)
//            -> GEMM wo (+x in place) -> RMSNorm -> GEMM w1|w3 with SwiGLU epilogue -> GEMM w2 (+x in place).
// Decode (<= 64 rows): wqkv / wo / w2 leave fp32 K-slices that RoPE-split and a fused residual-add + RMSNorm sum up.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "attention.hpp"
#include "ctx.hpp"
#include "decode.hpp"
#include "misc.hpp"
#include "norm.hpp"

struct cr_kv {
    cr_ctx* ctx;
    int n_seqs, max_tokens, gen_cap, layers;
    bf16* k;             // [layers][n_seqs][8][max_tokens][128]
    bf16* v;
    int32_t* d_len;      // [n_seqs] tokens cached
    int32_t* d_ngen;     // [n_seqs] ids generated
    int64_t* d_gen;      // [n_seqs][gen_cap]
    int32_t* d_seqs;     // [n_seqs] staging for the seqs[] of one decode call
    std::vector<int32_t> seqs_on_device;   // what d_seqs holds (decode skips the host-to-device copy of an unchanged list: a pageable copy
                                           // makes the host wait for the stream, which stops it from feeding a second stream)
    std::vector<int> len, ngen;
    // batched decode as a hipGraph (CR_DECODE_GRAPH=1): a step is ~290 short launches whose parameters only change with
    // the row count and the number of attention splits (positions, ids and cache lengths live in device memory), so the
    // launch sequence is captured once per (rows, splits, penalty) and replayed.  OFF by default: on ROCm 7.2 / MI355X the
    // replay measured 0.3 % slower than plain launches at 64 pages per step (6.86 vs 6.88 pages/s) and 5 % slower for a
    // single page (0.711 vs 0.678 s) -- the ~3 us gaps between dependent kernels are not launch overhead the graph removes
    struct DecodeGraph { int n, nsplit; float penalty; char* ws; uint64_t weight_gen; hipGraphExec_t exec; int warm; };
    std::vector<DecodeGraph> graphs;
};

namespace {

constexpr int D = 4096, HD = 128, NH = 32, NKV = 8, QKV = 6144;

// x[i] = tok_embeddings[id_i]; id_i = force[i] or the last generated id of seq_i
__global__ __launch_bounds__(256) void embed_rows_kernel(const bf16* __restrict__ table, const int64_t* __restrict__ force,
                                                         const int32_t* __restrict__ seqs, const int64_t* __restrict__ gen,
                                                         const int32_t* __restrict__ ngen, int gen_cap, bf16* __restrict__ x) {
    const int i = blockIdx.x;
    int64_t id;
    if (force) id = force[i];
    else { const int s = seqs[i]; id = gen[(int64_t)s * gen_cap + ngen[s] - 1]; }
    const bf16x8* src = (const bf16x8*)(table + id * D);
    bf16x8* dst = (bf16x8*)(x + (int64_t)i * D);
    dst[threadIdx.x] = src[threadIdx.x];
    dst[threadIdx.x + 256] = src[threadIdx.x + 256];
}

// splice: src_row[s] >= 0 -> table row; -1-r -> vit row r; -(1<<30)-r -> ref row r
__global__ __launch_bounds__(1024) void splice_index_kernel(const int64_t* __restrict__ ids, int S, int64_t img_id, int64_t ref_id,
                                                            int has_vit, int has_ref, int32_t* __restrict__ src_row,
                                                            int32_t* __restrict__ counts) {
    // one workgroup; running counts carried across chunks of 1024 ids
    __shared__ int s_img[1024], s_ref[1024];
    __shared__ int base_img, base_ref;
    const int tid = threadIdx.x;
    if (tid == 0) { base_img = 0; base_ref = 0; }
    __syncthreads();
    for (int s0 = 0; s0 < S; s0 += 1024) {
        const int s = s0 + tid;
        const int64_t id = s < S ? ids[s] : -1;
        const int is_img = (has_vit && id == img_id) ? 1 : 0;
        const int is_ref = (has_vit && has_ref && id == ref_id) ? 1 : 0;
        s_img[tid] = is_img; s_ref[tid] = is_ref;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = tid >= o ? s_img[tid - o] : 0, b = tid >= o ? s_ref[tid - o] : 0;
            __syncthreads();
            s_img[tid] += a; s_ref[tid] += b;
            __syncthreads();
        }
        if (s < S) {
            int v = (int)id;
            if (is_img) v = -1 - (base_img + s_img[tid] - 1);
            else if (is_ref) v = -(1 << 30) - (base_ref + s_ref[tid] - 1);
            src_row[s] = v;
        }
        __syncthreads();
        if (tid == 1023) { base_img += s_img[1023]; base_ref += s_ref[1023]; }
        __syncthreads();
    }
    if (tid == 0) { counts[0] = base_img; counts[1] = base_ref; }
}

__global__ __launch_bounds__(256) void splice_rows_kernel(const bf16* __restrict__ table, const bf16* __restrict__ vit,
                                                          const bf16* __restrict__ ref, const int32_t* __restrict__ src_row,
                                                          int n_vit, int n_ref, bf16* __restrict__ out) {
    const int s = blockIdx.x;
    const int v = src_row[s];
    const bf16* src;
    if (v >= 0) src = table + (int64_t)v * D;
    else if (v > -(1 << 30)) { const int r = -1 - v; if (r >= n_vit) return; src = vit + (int64_t)r * D; }
    else { const int r = -(1 << 30) - v; if (r >= n_ref) return; src = ref + (int64_t)r * D; }
    const bf16x8* sp = (const bf16x8*)src;
    bf16x8* dp = (bf16x8*)(out + (int64_t)s * D);
    dp[threadIdx.x] = sp[threadIdx.x];
    dp[threadIdx.x + 256] = sp[threadIdx.x + 256];
}

// dst[i] = src[idx[i]] (GATHER) or dst[idx[i]] = src[i]: rows of 4096 bf16 (the last prompt row of every page around the final layer's row-wise tail)
template <bool GATHER>
__global__ __launch_bounds__(256) void move_rows_kernel(const bf16* __restrict__ src, const int32_t* __restrict__ idx, bf16* __restrict__ dst) {
    const int i = blockIdx.x;
    const int64_t r = idx[i];
    const bf16x8* sp = (const bf16x8*)(src + (GATHER ? r : (int64_t)i) * D);
    bf16x8* dp = (bf16x8*)(dst + (GATHER ? (int64_t)i : r) * D);
    dp[threadIdx.x] = sp[threadIdx.x];
    dp[threadIdx.x + 256] = sp[threadIdx.x + 256];
}

// RoPE + split of the fused wqkv output (modeling_internlm2.py:359-388, 233-247).
// qkv row = [8 groups][4 q | k | v][128].  q_embed = bf16(bf16(q*cos) + bf16(rotate_half(q)*sin)), same for k.
// grid (rows, 8 groups), 128 threads: thread = (slot 0..7 [6 used], 16-B chunk 0..15).
__global__ __launch_bounds__(128) void rope_split_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ cosT,
                                                         const bf16* __restrict__ sinT, bf16* __restrict__ q_out,
                                                         bf16* __restrict__ kc, bf16* __restrict__ vc, int pos0, int seq0,
                                                         const int32_t* __restrict__ seqs, const int32_t* __restrict__ lens,
                                                         const int32_t* __restrict__ row_pos, int max_tokens,
                                                         const float* __restrict__ part, int splits) {
    // part != nullptr (decode): the wqkv output arrives as `splits` fp32 K-slices [s][rows][QKV]; their sum in slice
    // order, rounded once, is the bf16 linear output the reference rotates
    const int row = blockIdx.x, grp = blockIdx.y;
    const int slot = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (slot >= 6) return;
    const int seq = seqs ? seqs[row] : seq0;
    const int pos = row_pos ? row_pos[row] : (lens ? lens[seq] : pos0 + row);
    const int64_t col0 = (int64_t)(grp * 6 + slot) * HD;
    auto chunk = [&](int cc) -> bf16x8 {
        if (!part) return *(const bf16x8*)(qkv + (int64_t)row * QKV + col0 + cc * 8);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; s++) {
            const f32x4* pp = (const f32x4*)(part + ((int64_t)s * gridDim.x + row) * QKV + col0 + cc * 8);
            const f32x4 v0 = pp[0], v1 = pp[1];
#pragma unroll
            for (int e = 0; e < 4; e++) { a[e] += v0[e]; a[4 + e] += v1[e]; }
        }
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; e++) r[e] = f2bf(a[e]);
        return r;
    };
    const bf16x8 x = chunk(c);
    bf16x8 y;
    if (slot < 5) {
        const bf16x8 xp = chunk((c + 8) & 15);
        const bf16x8 cs = *(const bf16x8*)(cosT + (int64_t)pos * HD + c * 8);
        const bf16x8 sn = *(const bf16x8*)(sinT + (int64_t)pos * HD + c * 8);
        const float sign = c < 8 ? -1.0f : 1.0f;           // rotate_half: (-x2, x1)
#pragma unroll
        for (int e = 0; e < 8; e++)
            y[e] = f2bf(rbf(bf2f(x[e]) * bf2f(cs[e])) + rbf(sign * bf2f(xp[e]) * bf2f(sn[e])));
    } else {
        y = x;
    }
    bf16* dst;
    if (slot < 4) dst = q_out + (int64_t)row * D + (grp * 4 + slot) * HD;
    else {
        bf16* base = slot == 4 ? kc : vc;
        dst = base + (((int64_t)seq * NKV + grp) * max_tokens + pos) * HD;
    }
    *(bf16x8*)(dst + c * 8) = y;
}

// RepetitionPenaltyLogitsProcessor (published 4.45.2 semantics) + argmax (first max wins) + append the id.
// One workgroup of 1024 per row.
__global__ __launch_bounds__(1024) void pick_kernel(float* __restrict__ logits, int64_t ld, int vocab, float penalty,
                                                    int seq0, const int32_t* __restrict__ seqs, int64_t* __restrict__ gen,
                                                    int32_t* __restrict__ ngen, int32_t* __restrict__ lens, int gen_cap, int len_add) {
    __shared__ float bv[1024];
    __shared__ int bi[1024];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int seq = seqs ? seqs[row] : seq0;
    float* lg = logits + (int64_t)row * ld;
    const int ng = ngen[seq];
    int64_t* g = gen + (int64_t)seq * gen_cap;
    if (penalty != 1.0f && ng > 0) {
        // gather every score first, then scatter: duplicates of an id all see the ORIGINAL score
        float sv[4]; int64_t si[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int h = tid + k * 1024;
            si[k] = h < ng ? g[h] : -1;
            sv[k] = si[k] >= 0 ? lg[si[k]] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (si[k] >= 0) lg[si[k]] = sv[k] < 0.f ? sv[k] * penalty : sv[k] / penalty;
        __syncthreads();
    }
    float best = -INFINITY; int besti = 0x7fffffff;
    for (int c = tid; c < vocab; c += 1024) {
        const float v = lg[c];
        if (v > best) { best = v; besti = c; }
    }
    bv[tid] = best; bi[tid] = besti;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            const float v2 = bv[tid + o]; const int i2 = bi[tid + o];
            if (v2 > bv[tid] || (v2 == bv[tid] && i2 < bi[tid])) { bv[tid] = v2; bi[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (ng < gen_cap) { g[ng] = bi[0]; ngen[seq] = ng + 1; }
        lens[seq] += len_add;
    }
}

// w13 rows: [16q + e] = w1[8q + e], [16q + 8 + e] = w3[8q + e]   (pairs gate/up for the SwiGLU epilogue)
__global__ __launch_bounds__(256) void interleave8_kernel(const bf16* __restrict__ w1, const bf16* __restrict__ w3,
                                                          bf16* __restrict__ out, int ff) {
    const int r = blockIdx.x;                 // output row, 0 .. 2*ff
    const int q = r >> 4, e = r & 15;
    const bf16* src = (e < 8 ? w1 : w3) + (int64_t)(q * 8 + (e & 7)) * D;
    const bf16x8* sp = (const bf16x8*)src;
    bf16x8* dp = (bf16x8*)(out + (int64_t)r * D);
    dp[threadIdx.x] = sp[threadIdx.x];
    dp[threadIdx.x + 256] = sp[threadIdx.x + 256];
}

// One weight row -> e4m3 bytes + its scale: scale = max|w| / 448 (the largest e4m3 magnitude), q = round-to-nearest-even
// (w / scale); an all-zero row keeps scale 1.  One workgroup per row, K % 8 == 0.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16* __restrict__ w, int64_t ldw, int K, unsigned char* __restrict__ q,
                                                              float* __restrict__ scale) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const bf16* wr = w + r * ldw;
    float mx = 0.f;
    for (int k = tid * 8; k < K; k += 256 * 8) {
        const bf16x8 v = *(const bf16x8*)(wr + k);
#pragma unroll
        for (int e = 0; e < 8; e++) mx = fmaxf(mx, fabsf(bf2f(v[e])));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = mx > 0.f ? mx / 448.0f : 1.0f;
    const float inv = 1.0f / sc;
    if (tid == 0) scale[r] = sc;
    for (int k = tid * 8; k < K; k += 256 * 8) {
        const bf16x8 v = *(const bf16x8*)(wr + k);
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[0]) * inv, bf2f(v[1]) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[2]) * inv, bf2f(v[3]) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[4]) * inv, bf2f(v[5]) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[6]) * inv, bf2f(v[7]) * inv, hi, true);
        *(uint2*)(q + r * (int64_t)K + k) = make_uint2(lo, hi);
    }
}

// out[0] = max over rows of ||w_row||_2, out[1] = max |b| (positive floats order like their bit patterns: atomicMax on the bits)
__global__ __launch_bounds__(256) void row_norm_max_kernel(const bf16* __restrict__ w, int K, const bf16* __restrict__ b, float* __restrict__ out) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const bf16* wr = w + (int64_t)blockIdx.x * K;
    float ss = 0.f;
    for (int k = tid; k < K; k += 256) { const float v = bf2f(wr[k]); ss += v * v; }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        atomicMax((unsigned*)out, __float_as_uint(sqrtf(red[0] + red[1] + red[2] + red[3])));
        if (b) atomicMax((unsigned*)out + 1, __float_as_uint(fabsf(bf2f(b[blockIdx.x]))));
    }
}

int gemm(cr_ctx* c, int epi, const bf16* A, int64_t lda, const bf16* Wt, int64_t ldw, void* C, int64_t ldc, const bf16* res, int64_t ldr,
         int M, int N, int K, hipStream_t st, bool prefill_rows = false, const DevTensor* dl = nullptr, int dl_kind = 1) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = Wt; p.ldw = ldw; p.C = C; p.ldc = ldc; p.res = res; p.ldr = ldr; p.M = M; p.N = N; p.K = K;
    // decode (<= 64 rows): the weight-streaming kernels take the decode-layout copy where llm_finalize made one (a contiguous KiB per load instruction)
    if (dl && !prefill_rows && M <= 64) { p.W = (const bf16*)dl->ptr; p.wsw = dl_kind; }
    // a prompt row's result must not depend on what it is prefilled with: prompts of <= 64 rows in all stay on the tiled kernel, whose K
    // order is that of every longer prefill (the weight-streaming kernel the dispatcher would pick splits K over its waves)
    if (prefill_rows && M <= 64) p.kernel = 128;
    return ctx_gemm(c, epi, p, st);
}

int rms(const bf16* in, int64_t ld_in, bf16* out, const bf16* w, int64_t rows, float eps, hipStream_t st, float* row_scale = nullptr) {
    NormParams np{};
    np.in = in; np.ld_in = ld_in; np.out = out; np.ld_out = D; np.gamma = w; np.rows = rows; np.eps = eps;
    if (row_scale) { np.out8 = (unsigned char*)out; np.out8_scale = row_scale; }     // e4m3 rows in the first half of `out`
    return launch_rmsnorm(np, D, st);
}

// the same GEMM on the e4m3 copy of a weight (decode only): C = (A . W8^T) * wscale
int gemm8(cr_ctx* c, int epi, const bf16* A, int64_t lda, const DevTensor* w8, const DevTensor* ws, void* C, int64_t ldc, const bf16* res,
          int64_t ldr, int M, int N, int K, hipStream_t st, const DevTensor* dl = nullptr) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = (const bf16*)w8->ptr; p.ldw = K; p.C = C; p.ldc = ldc; p.res = res; p.ldr = ldr; p.M = M; p.N = N; p.K = K;
    p.w8 = 1; p.wscale = (const float*)ws->ptr;
    if (dl && M <= 64) { p.W = (const bf16*)dl->ptr; p.wsw = 1; }       // the e4m3 copy in its decode layout (cr_enable_fp8_decode): a contiguous KiB per load instruction
    return ctx_gemm(c, epi, p, st);
}

struct LayerW { const bf16 *an, *fn, *wqkv, *wo, *w13, *w2; const DevTensor *d_qkv, *d_o, *d_13, *d_2;      // d_*: decode-layout copies (llm_finalize), may be null
                const DevTensor *q_qkv, *s_qkv, *q_o, *s_o, *q_13, *s_13, *q_2, *s_2;
                const DevTensor *l_qkv, *l_o, *l_13, *l_2; };      // l_*: the e4m3 copies in their decode layout (cr_enable_fp8_decode), may be null

const DevTensor* opt(cr_ctx* c, const std::string& name) {
    auto it = c->w.find(name);
    return it == c->w.end() ? nullptr : &it->second;
}

int layer_weights(cr_ctx* c, int l, LayerW& w) {
    const std::string p = "language_model.model.layers." + std::to_string(l) + ".";
    w.an = W(c, p + "attention_norm.weight"); w.fn = W(c, p + "ffn_norm.weight");
    w.wqkv = W(c, p + "attention.wqkv.weight"); w.wo = W(c, p + "attention.wo.weight");
    w.w13 = W(c, "derived.w13." + std::to_string(l)); w.w2 = W(c, p + "feed_forward.w2.weight");
    w.d_qkv = opt(c, "declayout." + p + "attention.wqkv.weight"); w.d_o = opt(c, "declayout." + p + "attention.wo.weight");
    w.d_13 = opt(c, "declayout.derived.w13." + std::to_string(l)); w.d_2 = opt(c, "declayout." + p + "feed_forward.w2.weight");
    w.q_qkv = opt(c, "fp8." + p + "attention.wqkv.weight"); w.s_qkv = opt(c, "fp8s." + p + "attention.wqkv.weight");
    w.q_o = opt(c, "fp8." + p + "attention.wo.weight"); w.s_o = opt(c, "fp8s." + p + "attention.wo.weight");
    w.q_13 = opt(c, "fp8.derived.w13." + std::to_string(l)); w.s_13 = opt(c, "fp8s.derived.w13." + std::to_string(l));
    w.q_2 = opt(c, "fp8." + p + "feed_forward.w2.weight"); w.s_2 = opt(c, "fp8s." + p + "feed_forward.w2.weight");
    w.l_qkv = opt(c, "fp8dl." + p + "attention.wqkv.weight"); w.l_o = opt(c, "fp8dl." + p + "attention.wo.weight");
    w.l_13 = opt(c, "fp8dl.derived.w13." + std::to_string(l)); w.l_2 = opt(c, "fp8dl." + p + "feed_forward.w2.weight");
    return (w.an && w.fn && w.wqkv && w.wo && w.w13 && w.w2) ? CR_OK : CR_ERR_STATE;
}

struct Segment { int seq, row0, S, pos0; };     // a page's rows inside a batched prefill

// The decoder stack over M rows (prefill: M = all prompt rows of the pages in `segs`; decode: M = n sequences, one row each).
// d_last / d_seg_last (prefill): the pages' last rows and, as attention segments, the rows that share a wave with them -- see the final layer below.
int run_layers(cr_ctx* c, cr_kv* kv, bf16* x, int M, bool decode, const std::vector<Segment>& segs, const int32_t* d_row_seq,
               const int32_t* d_row_pos, const int32_t* d_seqs, int nsplit, hipStream_t st, const int32_t* d_seg = nullptr,
               const int32_t* d_last = nullptr, const int32_t* d_seg_last = nullptr) {
    const int ff = (int)WT(c, "derived.w13.0")->shape[0] / 2;
    Arena ar(c->ws);
    float* part = decode ? ar.take<float>(attn_split_ws_floats(M, NKV, NH / NKV, nsplit, HD)) : nullptr;
    bf16* h = ar.take<bf16>((size_t)M * D);
    bf16* qkv = ar.take<bf16>((size_t)M * QKV);
    bf16* q = ar.take<bf16>((size_t)M * D);
    bf16* ao = ar.take<bf16>((size_t)M * D);
    bf16* act = ar.take<bf16>((size_t)M * ff);
    float* hs = ar.take<float>((size_t)M);          // fp8 matrix-core path: one scale per normalised row
    const bool any8 = !decode && c->fp8_mfma && M >= 256;
    unsigned char* a8 = any8 ? ar.take<unsigned char>((size_t)M * ff) : nullptr;     // e4m3 rows of the attention output / the SwiGLU output
    // Prefill, FINAL layer: only each page's last row is ever read again (the LM head, :1081-1082 + _sample; the reference computes every row).  K / V of
    // every row still go to the cache, so the norm, wqkv and RoPE + split run on all rows; attention, wo, the second norm, w1|w3 and w2 run on the n last rows
    // only -- 88 % of that layer's linear work.  Bit-exact: the tiled kernels give a row the same sum whatever it is batched with (pinned, as for short
    // prompts), and the attention launch keeps a last row in the company of the rows that share its WAVE in the full launch (the kernel's deferred-rescale vote
    // is wave-wide): segment = {first row of that 32-row group, rows up to the last, their first position}.  CR_PREFILL_LAST_ROWS=0: every row (A/B aid).
    const int n_pages = (int)segs.size();
    const bool last_rows_only = !decode && c->prefill_last_rows && d_last && d_seg_last && !any8 && !c->probe_dst && n_pages > 0;
    bf16 *xl = nullptr, *aol = nullptr, *hl2 = nullptr, *actl = nullptr;
    if (last_rows_only) {
        xl = ar.take<bf16>((size_t)n_pages * D); aol = ar.take<bf16>((size_t)n_pages * D); hl2 = ar.take<bf16>((size_t)n_pages * D);
        actl = ar.take<bf16>((size_t)n_pages * ff);
    }
    const bf16 *cosT = W(c, "rope.cos"), *sinT = W(c, "rope.sin");
    if (!cosT || !sinT) return CR_ERR_STATE;
    const int64_t per_layer = (int64_t)kv->n_seqs * NKV * kv->max_tokens * HD;
    // decode (M <= 64): wqkv, wo and w2 run as K-sliced partial-sum GEMMs (gemm_skinny.hip: tall workgroups re-read X
    // once per 64 weight rows); the slices are summed by the kernel that consumes the result anyway -- RoPE/split for
    // wqkv, the residual add fused with the NEXT RMSNorm for wo and w2 -- in slice order, so a row's result is the same
    // whatever it is batched with.
    const int s_qkv = decode ? gemm_partial_splits(QKV, D) : 0, s_o = decode ? gemm_partial_splits(D, D) : 0,
              s_2 = decode ? gemm_partial_splits(D, ff) : 0;
    const bool sliced = decode && M <= 64 && s_qkv > 0 && s_o > 0 && s_2 > 0 && !c->no_sliced_decode;
    float* pbuf = sliced ? ar.take<float>((size_t)std::max(s_qkv * QKV, std::max(s_o, s_2) * D) * M) : nullptr;
    // small batches (<= DECODE_FUSED_MAX_ROWS = 8 rows): gemm_decode.hip folds the norms, RoPE + split and the residual adds into the five GEMMs -- six launches
    // per layer instead of nine, every sum in the order of the kernels below (a row's bits do not depend on its batch): its virtual slices are
    // compiled in, so the path is only taken while the K-sliced kernels' geometry (cost model or CR_PARTIAL_GEOM) is the one they reproduce
    const bool fused = sliced && c->fused_decode && !c->fp8_decode && decode_fused_supported(M, ff) &&
                       s_qkv == DEC_SLICES_WQKV && s_o == DEC_SLICES_WO && s_2 == DEC_SLICES_W2;
    // cr_llm_hidden_probe (parity tooling): rows [row0, row0 + rows) of the residual stream before layer 0 and after every layer
    auto probe = [&](int slot) -> int {
        if (!c->probe_dst || decode || c->probe_row0 + c->probe_rows > M) return CR_OK;
        CR_HIP(hipMemcpyAsync(c->probe_dst + (size_t)slot * c->probe_rows * D, x + (size_t)c->probe_row0 * D, (size_t)c->probe_rows * D * 2,
                              hipMemcpyDeviceToDevice, st));
        return CR_OK;
    };
    CR_TRY(probe(0));
    for (int l = 0; l < c->d.llm_layers; l++) {
        LayerW w;
        CR_TRY(layer_weights(c, l, w));
        bf16* kc = kv->k + l * per_layer;
        bf16* vc = kv->v + l * per_layer;
        // prefill on the fp8 matrix-core path: both norms emit e4m3 rows + scales, wqkv and w1|w3 take them (gemm256 F8)
        if (fused) {
            DecodeGemmParams dp{};
            dp.M = M; dp.eps = c->d.rms_eps;
            dp.W = w.wqkv; dp.ldw = D; dp.N = QKV; dp.K = D; dp.xres = x; dp.gamma = w.an;
            if (w.d_qkv) { dp.W = (const bf16*)w.d_qkv->ptr; dp.swizzled = 1; }
            dp.cosT = cosT; dp.sinT = sinT; dp.q_out = q; dp.kc = kc; dp.vc = vc; dp.seqs = d_seqs; dp.lens = kv->d_len; dp.max_tokens = kv->max_tokens;
            CR_TRY(launch_decode_gemm(DEC_WQKV, dp, st));
            AttnParams ap{};
            ap.K = kc; ap.V = vc; ap.Q = q; ap.O = ao;
            ap.k_bs = ap.v_bs = (int64_t)NKV * kv->max_tokens * HD; ap.k_rs = ap.v_rs = HD; ap.k_hs = ap.v_hs = (int64_t)kv->max_tokens * HD;
            ap.q_prescale = 1.0f; ap.s_div = 11.313708498984761f;
            ap.q_bs = D; ap.q_rs = HD; ap.q_hs = 4 * HD; ap.o_bs = D; ap.o_rs = HD; ap.o_hs = 4 * HD;
            ap.B = M; ap.H = NKV; ap.Sq = NH / NKV; ap.Sk = 0; ap.kv_group = 1; ap.q_pos0 = 0;
            ap.seq_map = d_seqs; ap.sk_arr = kv->d_len; ap.sk_add = 1;
            ap.nsplit = nsplit; ap.part_ml = part; ap.part_o = part + (size_t)M * NKV * nsplit * (NH / NKV) * 2;
            if (launch_flash_attn_split(ap, HD, st) != CR_OK) return cr_fail(CR_ERR_HIP, "decode attention launch failed");
            DecodeGemmParams dq{};
            dq.M = M; dq.eps = c->d.rms_eps;
            dq.W = w.wo; dq.ldw = D; dq.N = D; dq.K = D; dq.X = ao; dq.ldx = D; dq.xio = x;
            if (w.d_o) { dq.W = (const bf16*)w.d_o->ptr; dq.swizzled = 1; }
            CR_TRY(launch_decode_gemm(DEC_WO, dq, st));
            dq = DecodeGemmParams{};
            dq.M = M; dq.eps = c->d.rms_eps;
            dq.W = w.w13; dq.ldw = D; dq.N = 2 * ff; dq.K = D; dq.xres = x; dq.gamma = w.fn; dq.C = act; dq.ldc = ff;
            if (w.d_13) { dq.W = (const bf16*)w.d_13->ptr; dq.swizzled = 1; }
            CR_TRY(launch_decode_gemm(DEC_W13, dq, st));
            dq = DecodeGemmParams{};
            dq.M = M; dq.eps = c->d.rms_eps;
            dq.W = w.w2; dq.ldw = ff; dq.N = D; dq.K = ff; dq.X = act; dq.ldx = ff; dq.xio = x;
            if (w.d_2) { dq.W = (const bf16*)w.d_2->ptr; dq.swizzled = 1; }
            CR_TRY(launch_decode_gemm(DEC_W2, dq, st));
            continue;
        }
        const bool m8 = any8 && w.q_qkv && w.s_qkv && w.q_13 && w.s_13;
        const bool m8all = m8 && c->fp8_mfma >= 2 && w.q_o && w.s_o && w.q_2 && w.s_2;      // level 2: wo and w2 as well (their inputs take a quantiser pass)
        if (!sliced || l == 0) CR_TRY(rms(x, D, h, w.an, M, c->d.rms_eps, st, m8 ? hs : nullptr));
        const bool f8 = decode && M <= 64 && c->fp8_decode && w.q_qkv && w.s_qkv && w.q_o && w.s_o && w.q_13 && w.s_13 && w.q_2 && w.s_2;
        if (m8) CR_TRY(ctx_gemm_f8(c, EPI_STORE, h, hs, w.q_qkv, w.s_qkv, qkv, QKV, nullptr, M, QKV, D, st));
        else if (sliced) CR_TRY(f8 ? gemm8(c, EPI_PARTIAL, h, D, w.q_qkv, w.s_qkv, pbuf, QKV, nullptr, 0, M, QKV, D, st, w.l_qkv)
                               : gemm(c, EPI_PARTIAL, h, D, w.wqkv, D, pbuf, QKV, nullptr, 0, M, QKV, D, st, false, w.d_qkv, 2));
        else CR_TRY(f8 ? gemm8(c, EPI_STORE, h, D, w.q_qkv, w.s_qkv, qkv, QKV, nullptr, 0, M, QKV, D, st, w.l_qkv)
                        : gemm(c, EPI_STORE, h, D, w.wqkv, D, qkv, QKV, nullptr, 0, M, QKV, D, st, !decode));
        AttnParams ap{};
        ap.K = kc; ap.V = vc; ap.Q = q; ap.O = ao;
        ap.k_bs = ap.v_bs = (int64_t)NKV * kv->max_tokens * HD; ap.k_rs = ap.v_rs = HD; ap.k_hs = ap.v_hs = (int64_t)kv->max_tokens * HD;
        ap.q_prescale = 1.0f; ap.s_div = 11.313708498984761f;      // math.sqrt(128)
        // decode, 9..64 rows: RoPE + split + the new token's cache row are folded into the attention kernel (attention_decode.hip, FOLD) where it qualifies
        bool rope_folded = false;
        if (sliced && c->fold_rope) {
            AttnParams fp = ap;
            fp.q_bs = D; fp.q_rs = HD; fp.q_hs = 4 * HD; fp.B = M; fp.H = NKV; fp.Sq = NH / NKV; fp.kv_group = 1; fp.sk_arr = kv->d_len; fp.sk_add = 1;
            fp.nsplit = nsplit; fp.part_ml = part; fp.part_o = part + (size_t)M * NKV * nsplit * (NH / NKV) * 2;
            fp.qkv_part = pbuf; fp.qkv_splits = s_qkv; fp.qkv_ld = QKV; fp.rope_cos = cosT; fp.rope_sin = sinT;
            rope_folded = decode_attn_fold_supported(fp, HD);
        }
        if (!rope_folded)
            hipLaunchKernelGGL(rope_split_kernel, dim3(M, NKV), dim3(128), 0, st, qkv, cosT, sinT, q, kc, vc, 0, 0,
                               decode ? d_seqs : d_row_seq, decode ? kv->d_len : nullptr, decode ? nullptr : d_row_pos, kv->max_tokens,
                               (const float*)(sliced ? pbuf : nullptr), s_qkv);
        if (last_rows_only && l + 1 == c->d.llm_layers) {
            ap.q_bs = 0; ap.q_rs = D; ap.q_hs = HD; ap.o_bs = 0; ap.o_rs = D; ap.o_hs = HD;
            ap.H = NH; ap.kv_group = NH / NKV; ap.B = n_pages; ap.seg = d_seg_last; ap.Sq = 32;
            if (launch_flash_attn(ap, HD, true, st) != CR_OK) return cr_fail(CR_ERR_HIP, "prefill attention launch failed");
            hipLaunchKernelGGL(move_rows_kernel<true>, dim3(n_pages), dim3(256), 0, st, ao, d_last, aol);
            hipLaunchKernelGGL(move_rows_kernel<true>, dim3(n_pages), dim3(256), 0, st, x, d_last, xl);
            // (rows <= 64 are pinned to the tiled kernel by `prefill_rows`; more pages than that take it by the dispatcher's own rule)
            CR_TRY(gemm(c, EPI_RES, aol, D, w.wo, D, xl, D, xl, D, n_pages, D, D, st, true));
            CR_TRY(rms(xl, D, hl2, w.fn, n_pages, c->d.rms_eps, st));
            CR_TRY(gemm(c, EPI_SWIGLU, hl2, D, w.w13, D, actl, ff, nullptr, 0, n_pages, 2 * ff, D, st, true));
            CR_TRY(gemm(c, EPI_RES, actl, ff, w.w2, ff, xl, D, xl, D, n_pages, D, ff, st, true));
            hipLaunchKernelGGL(move_rows_kernel<false>, dim3(n_pages), dim3(256), 0, st, xl, d_last, x);
            continue;
        }
        if (!decode) {
            ap.q_bs = 0; ap.q_rs = D; ap.q_hs = HD; ap.o_bs = 0; ap.o_rs = D; ap.o_hs = HD;
            ap.B = 1; ap.H = NH; ap.kv_group = NH / NKV;
            // causal attention is per page, the linear layers see all pages at once; the pages' query blocks go out as ONE launch
            // (segments: first row, rows, first position, cache slot), longest blocks first: a launch per page left its 800
            // workgroups of 2..50 key tiles each a ragged tail on 512 slots
            ap.B = (int)segs.size(); ap.seg = d_seg; ap.Sq = 0;
            for (const Segment& sg : segs) ap.Sq = sg.S > ap.Sq ? sg.S : ap.Sq;
            if (launch_flash_attn(ap, HD, true, st) != CR_OK) return cr_fail(CR_ERR_HIP, "prefill attention launch failed");
        } else {
            // rows = the 4 query heads of one KV group, all at the same position: no mask needed
            ap.q_bs = D; ap.q_rs = HD; ap.q_hs = 4 * HD; ap.o_bs = D; ap.o_rs = HD; ap.o_hs = 4 * HD;
            ap.B = M; ap.H = NKV; ap.Sq = NH / NKV; ap.Sk = 0; ap.kv_group = 1; ap.q_pos0 = 0;
            ap.seq_map = d_seqs; ap.sk_arr = kv->d_len; ap.sk_add = 1;
            ap.nsplit = nsplit; ap.part_ml = part; ap.part_o = part + (size_t)M * NKV * nsplit * (NH / NKV) * 2;
            if (rope_folded) { ap.qkv_part = pbuf; ap.qkv_splits = s_qkv; ap.qkv_ld = QKV; ap.rope_cos = cosT; ap.rope_sin = sinT; }
            if (launch_flash_attn_split(ap, HD, st) != CR_OK) return cr_fail(CR_ERR_HIP, "decode attention launch failed");
        }
        if (sliced) {
            CR_TRY(f8 ? gemm8(c, EPI_PARTIAL, ao, D, w.q_o, w.s_o, pbuf, D, nullptr, 0, M, D, D, st, w.l_o)
                      : gemm(c, EPI_PARTIAL, ao, D, w.wo, D, pbuf, D, nullptr, 0, M, D, D, st, false, w.d_o));
            CR_TRY(launch_add_rmsnorm(x, pbuf, s_o, M, w.fn, h, c->d.rms_eps, st));
        } else {
            if (m8all) {    // wo and w2 take activations no norm produced: one quantiser pass each (row maximum, then the e4m3 row)
                hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)M), dim3(256), 0, st, ao, (int64_t)D, D, a8, hs);
                CR_TRY(ctx_gemm_f8(c, EPI_RES, a8, hs, w.q_o, w.s_o, x, D, nullptr, M, D, D, st, x, D));
            } else {
                CR_TRY(f8 ? gemm8(c, EPI_RES, ao, D, w.q_o, w.s_o, x, D, x, D, M, D, D, st, w.l_o)
                          : gemm(c, EPI_RES, ao, D, w.wo, D, x, D, x, D, M, D, D, st, !decode));
            }
            CR_TRY(rms(x, D, h, w.fn, M, c->d.rms_eps, st, m8 ? hs : nullptr));
        }
        if (m8) CR_TRY(ctx_gemm_f8(c, EPI_SWIGLU, h, hs, w.q_13, w.s_13, act, ff, nullptr, M, 2 * ff, D, st));
        else CR_TRY(f8 ? gemm8(c, EPI_SWIGLU, h, D, w.q_13, w.s_13, act, ff, nullptr, 0, M, 2 * ff, D, st, w.l_13)
                       : gemm(c, EPI_SWIGLU, h, D, w.w13, D, act, ff, nullptr, 0, M, 2 * ff, D, st, !decode, decode ? w.d_13 : nullptr));
        if (sliced) {
            CR_TRY(f8 ? gemm8(c, EPI_PARTIAL, act, ff, w.q_2, w.s_2, pbuf, D, nullptr, 0, M, D, ff, st, w.l_2)
                      : gemm(c, EPI_PARTIAL, act, ff, w.w2, ff, pbuf, D, nullptr, 0, M, D, ff, st, false, w.d_2));
            const bf16* next_norm = nullptr;             // the last layer's sum only lands in x: the caller norms what it needs
            if (l + 1 < c->d.llm_layers) {
                LayerW wn;
                CR_TRY(layer_weights(c, l + 1, wn));
                next_norm = wn.an;
            }
            CR_TRY(launch_add_rmsnorm(x, pbuf, s_2, M, next_norm, h, c->d.rms_eps, st));
        } else if (m8all) {
            hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)M), dim3(256), 0, st, act, (int64_t)ff, ff, a8, hs);
            CR_TRY(ctx_gemm_f8(c, EPI_RES, a8, hs, w.q_2, w.s_2, x, D, nullptr, M, D, ff, st, x, D));
        } else {
            CR_TRY(f8 ? gemm8(c, EPI_RES, act, ff, w.q_2, w.s_2, x, D, x, D, M, D, ff, st, w.l_2)
                      : gemm(c, EPI_RES, act, ff, w.w2, ff, x, D, x, D, M, D, ff, st, !decode));
        }
        CR_TRY(probe(l + 1));
    }
    CR_HIP(hipGetLastError());
    return CR_OK;
}

size_t layers_ws(cr_ctx* c, int M, int nsplit = 0, int n_pages = 0) {
    const size_t ff = (size_t)WT(c, "derived.w13.0")->shape[0] / 2;
    const size_t last_rows = (size_t)n_pages * (3 * D + ff) * 2 + 4 * 256;      // the final layer's compact rows (prefill)
    const size_t sliced = nsplit > 0 && M <= 64 ? (size_t)8 * QKV * M * 4 : 0;       // K-slice partial sums of the decode GEMMs
    return ((size_t)M * D * 4 + (size_t)M * QKV + (size_t)M * ff) * 2 + attn_split_ws_floats(M, NKV, NH / NKV, nsplit, HD) * 4 + sliced + (size_t)M * 4 + (size_t)M * ff + last_rows + 8192;
}

}  // namespace

int build_fp8_copy(cr_ctx* c, const std::string& nm, int k_multiple, hipStream_t st) {
    const DevTensor* src = WT(c, nm);
    if (!src) return CR_ERR_STATE;
    const int64_t N = src->shape[0], K = src->shape[1];
    if ((K % k_multiple) != 0) return cr_fail(CR_ERR_ARG, "fp8 copy: %s has K = %lld, not a multiple of %d", nm.c_str(), (long long)K, k_multiple);
    auto have = c->w.find("fp8." + nm);
    if (have != c->w.end() && have->second.shape == src->shape && c->w.count("fp8s." + nm)) return CR_OK;     // built already for these weights
    DevTensor q, s;
    q.dtype = CR_U8; q.shape = src->shape; q.bytes = (size_t)N * K;
    s.dtype = CR_F32; s.shape = {N}; s.bytes = (size_t)N * 4;
    if (hipMalloc(&q.ptr, q.bytes) != hipSuccess) return cr_fail(CR_ERR_NOMEM, "fp8 copy: %s", nm.c_str());
    if (hipMalloc(&s.ptr, s.bytes) != hipSuccess) { hipFree(q.ptr); return cr_fail(CR_ERR_NOMEM, "fp8 copy: %s", nm.c_str()); }
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)N), dim3(256), 0, st, (const bf16*)src->ptr, K, (int)K, (unsigned char*)q.ptr, (float*)s.ptr);
    for (const char* pre : {"fp8.", "fp8s."}) {
        auto old = c->w.find(pre + nm);
        if (old != c->w.end()) { hipFree(old->second.ptr); c->w.erase(old); }
    }
    c->w["fp8." + nm] = q;
    c->w["fp8s." + nm] = s;
    return CR_OK;
}

int ctx_gemm_f8(cr_ctx* c, int epi, const void* a8, const float* ascale, const DevTensor* w8, const DevTensor* ws, void* C, int64_t ldc,
                const bf16* bias, int M, int N, int K, hipStream_t st, const bf16* res, int64_t ldr) {
    GemmParams p{};
    p.res = res; p.ldr = ldr;
    p.A = (const bf16*)a8; p.lda = K; p.W = (const bf16*)w8->ptr; p.ldw = K; p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N; p.K = K;
    p.w8 = 1; p.wscale = (const float*)ws->ptr; p.a8 = 1; p.ascale = ascale;
    return ctx_gemm(c, epi, p, st);
}

int llm_finalize(cr_ctx* c, hipStream_t st) {
    for (int l = 0; l < c->d.llm_layers; l++) {
        const std::string p = "language_model.model.layers." + std::to_string(l) + ".feed_forward.";
        auto i1 = c->w.find(p + "w1.weight"), i3 = c->w.find(p + "w3.weight");
        const std::string dn = "derived.w13." + std::to_string(l);
        if (i1 == c->w.end() || i3 == c->w.end()) {
            // one of the pair reloaded without the other: the interleaved tensor cannot be rebuilt (the other original was released)
            if (i1 != c->w.end() || i3 != c->w.end()) return cr_fail(CR_ERR_STATE, "layer %d: reload feed_forward.w1 and w3 together", l);
            if (c->w.count(dn)) continue;          // already derived, originals released
            return cr_fail(CR_ERR_STATE, "layer %d: w1/w3 missing", l);
        }
        const int ff = (int)i1->second.shape[0];
        if (i1->second.shape[1] != D || i3->second.shape != i1->second.shape || (ff % 8))
            return cr_fail(CR_ERR_ARG, "layer %d: w1/w3 must be [ff,4096] with ff %% 8 == 0", l);
        DevTensor t;
        t.dtype = CR_BF16; t.shape = {2 * ff, D}; t.bytes = (size_t)2 * ff * D * 2;
        auto old = c->w.find(dn);
        if (old != c->w.end()) { hipFree(old->second.ptr); c->w.erase(old); }
        // the e4m3 copies derived from the interleaved tensor carry ITS name, not w1's / w3's: cr_load_weight's invalidation (by the reloaded tensor's own name)
        // cannot see them (round-5 advice).  They go with the tensor they were made from, and the fp8 options switch off until cr_enable_fp8_* rebuilds them.
        for (const char* pre : {"fp8.", "fp8s.", "fp8b.", "fp8dl."}) {
            auto d8 = c->w.find(std::string(pre) + dn);
            if (d8 == c->w.end()) continue;
            hipFree(d8->second.ptr);
            c->w.erase(d8);
            c->fp8_decode = false; c->fp8_mfma = 0;
        }
        CR_HIP(hipMalloc(&t.ptr, t.bytes));
        hipLaunchKernelGGL(interleave8_kernel, dim3(2 * ff), dim3(256), 0, st, (const bf16*)i1->second.ptr,
                           (const bf16*)i3->second.ptr, (bf16*)t.ptr, ff);
        CR_HIP(hipStreamSynchronize(st));
        hipFree(i1->second.ptr); hipFree(i3->second.ptr);      // the interleaved copy replaces them
        c->w.erase(i1); c->w.erase(c->w.find(p + "w3.weight"));
        c->w[dn] = t;
    }
    // Decode-layout copies of the five weight streams of a small-batch decode step (gemm_decode.hip: one contiguous KiB per load instruction instead of
    // 16 rows x 64 bytes; wqkv 16.7 -> 11.1 us, w1|w3 43.1 -> 37.5, w2 26.0 -> 20.5, LM head 130 -> 116 at one row, the same bits): +15.9 GB on
    // InternLM2.5-7B, taken from the 288 GB only if the shapes are the ones the kernels are built for.  CR_DECODE_LAYOUT=0: none (A/B aid); a failed allocation
    // leaves the decode path on the nn.Linear layout.
    {
        for (auto it = c->w.begin(); it != c->w.end();) {         // copies of an earlier cr_finalize: the weights may have been reloaded
            if (it->first.rfind("declayout.", 0) == 0) { hipFree(it->second.ptr); it = c->w.erase(it); } else ++it;
        }
        const char* e = getenv("CR_DECODE_LAYOUT");
        const DevTensor* w13_0 = WT(c, "derived.w13.0");
        const bool want = !(e && atoi(e) == 0) && w13_0 && decode_fused_supported(1, (int)w13_0->shape[0] / 2);
        auto derive = [&](int which, const std::string& nm, int64_t N, int64_t K) -> bool {
            const DevTensor* src = WT(c, nm);
            if (!src || src->shape.size() != 2 || src->shape[0] != N || src->shape[1] != K) return false;
            DevTensor t;
            t.dtype = CR_BF16; t.shape = {(N + 15) / 16 * 16, K}; t.bytes = decode_swizzled_bytes((int)N, (int)K);
            if (hipMalloc(&t.ptr, t.bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (decode_swizzle_weight(which, (const bf16*)src->ptr, K, (int)N, (int)K, (bf16*)t.ptr, st) != CR_OK) { hipFree(t.ptr); return false; }
            c->w["declayout." + nm] = t;
            return true;
        };
        bool ok = want;
        for (int l = 0; ok && l < c->d.llm_layers; l++) {
            const std::string p = "language_model.model.layers." + std::to_string(l) + ".";
            const int64_t ff = w13_0->shape[0] / 2;
            ok = derive(DEC_WQKV, p + "attention.wqkv.weight", QKV, D) && derive(DEC_WO, p + "attention.wo.weight", D, D) &&
                 derive(DEC_W13, "derived.w13." + std::to_string(l), 2 * ff, D) && derive(DEC_W2, p + "feed_forward.w2.weight", D, ff);
        }
        if (ok) {
            const DevTensor* ow = WT(c, "language_model.output.weight");
            if (ow && ow->shape.size() == 2 && ow->shape[1] == D) ok = derive(DEC_HEAD, "language_model.output.weight", ow->shape[0], D);
        }
        CR_HIP(hipStreamSynchronize(st));
        if (want && !ok) {
            // all or nothing (round-4 advice): a partial set would leave the device nearly full and the decode path on two layouts
            size_t freed = 0;
            for (auto it = c->w.begin(); it != c->w.end();) {
                if (it->first.rfind("declayout.", 0) == 0) { freed += it->second.bytes; hipFree(it->second.ptr); it = c->w.erase(it); } else ++it;
            }
            fprintf(stderr, "[callireader_hip] no room (or not the shapes) for the decode-layout copies of the LLM weights: %zu MB released, decode streams the nn.Linear layout\n",
                    freed >> 20);
        }
    }
    return CR_OK;
}

extern "C" {

int cr_op_quantize_fp8(const void* w, int64_t ldw, int N, int K, void* q, float* scale, void* stream) {
    if (!w || !q || !scale || N <= 0 || K <= 0 || (K & 7) || (ldw & 7)) return cr_fail(CR_ERR_ARG, "cr_op_quantize_fp8: bad argument");
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16*)w, ldw, K, (unsigned char*)q, scale);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

int cr_enable_fp8_mfma(cr_ctx* c, int enable, void* stream) {
    if (!c) return cr_fail(CR_ERR_ARG, "cr_enable_fp8_mfma: null context");
    if (!enable) { if (c->fp8_mfma) cr_bump_gen(c); c->fp8_mfma = 0; return CR_OK; }      // captured decode graphs are keyed on weight_gen
    if (c->borrowed) return cr_fail(CR_ERR_STATE, "cr_enable_fp8_mfma: enable it on the context that owns the weights, then share again");
    if (!c->finalized) return cr_fail(CR_ERR_STATE, "cr_enable_fp8_mfma: call cr_finalize first");
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<std::string> names;
    // the linears whose input is a norm's output: that kernel emits the e4m3 row and its scale for free
    for (int l = 0; l < c->d.vit_layers; l++) {
        const std::string p = "vision_model.encoder.layers." + std::to_string(l) + ".";
        if (!c->w.count(p + "attn.qkv.weight")) continue;
        names.push_back(p + "attn.qkv.weight"); names.push_back(p + "mlp.fc1.weight"); names.push_back(p + "mlp.fc2.weight");
        // fc1's output goes to fc2 as e4m3 straight from fc1's epilogue (EPI_GELU_Q8); its row scale is a bound LayerNorm 2 derives:
        // "fp8b.<fc1>" = {max row norm of W1, max |b1|}
        const DevTensor *w1 = WT(c, p + "mlp.fc1.weight"), *b1 = WT(c, p + "mlp.fc1.bias");
        if (!w1 || !b1) return CR_ERR_STATE;
        if (!c->w.count("fp8b." + p + "mlp.fc1.weight")) {
            DevTensor t;
            t.dtype = CR_F32; t.shape = {2}; t.bytes = 8;
            if (hipMalloc(&t.ptr, 8) != hipSuccess) return cr_fail(CR_ERR_NOMEM, "cr_enable_fp8_mfma: bound of layer %d", l);
            CR_HIP(hipMemsetAsync(t.ptr, 0, 8, st));
            hipLaunchKernelGGL(row_norm_max_kernel, dim3((unsigned)w1->shape[0]), dim3(256), 0, st, (const bf16*)w1->ptr, (int)w1->shape[1], (const bf16*)b1->ptr, (float*)t.ptr);
            c->w["fp8b." + p + "mlp.fc1.weight"] = t;
        }
    }
    if (c->w.count("mlp1.1.weight")) names.push_back("mlp1.1.weight");
    for (int l = 0; l < c->d.llm_layers; l++) {
        const std::string p = "language_model.model.layers." + std::to_string(l) + ".";
        if (c->w.count(p + "attention.wqkv.weight")) {
            names.push_back(p + "attention.wqkv.weight"); names.push_back("derived.w13." + std::to_string(l));
            names.push_back(p + "attention.wo.weight"); names.push_back(p + "feed_forward.w2.weight");     // their inputs take a quantiser pass
        }
    }
    for (const std::string& nm : names) CR_TRY(build_fp8_copy(c, nm, 256, st));
    CR_HIP(hipGetLastError());
    CR_HIP(hipStreamSynchronize(st));
    c->fp8_mfma = enable >= 2 ? 2 : 1;
    cr_bump_gen(c);
    return CR_OK;
}

int cr_enable_fp8_decode(cr_ctx* c, int enable, void* stream) {
    if (!c) return cr_fail(CR_ERR_ARG, "cr_enable_fp8_decode: null context");
    if (!enable) { if (c->fp8_decode) cr_bump_gen(c); c->fp8_decode = false; return CR_OK; }
    if (c->borrowed) return cr_fail(CR_ERR_STATE, "cr_enable_fp8_decode: enable it on the context that owns the weights, then share again");
    if (!c->finalized) return cr_fail(CR_ERR_STATE, "cr_enable_fp8_decode: call cr_finalize first");
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<std::string> names;
    for (int l = 0; l < c->d.llm_layers; l++) {
        const std::string p = "language_model.model.layers." + std::to_string(l) + ".";
        names.push_back(p + "attention.wqkv.weight"); names.push_back(p + "attention.wo.weight");
        names.push_back("derived.w13." + std::to_string(l)); names.push_back(p + "feed_forward.w2.weight");
    }
    names.push_back("language_model.output.weight");
    for (const std::string& nm : names) CR_TRY(build_fp8_copy(c, nm, 512, st));
    // ... and each of them once more in the decode layout (gemm_decode.hip: decode_swizzle8_kernel; + 7.7 GB on InternLM2.5-7B): without it the e4m3-weight
    // decode LOST to the bf16 decode on ITS decode layout (round 4: 9.76 against 9.23 ms per 64-row step).  All or nothing; CR_DECODE_LAYOUT=0: none.
    {
        const char* e = getenv("CR_DECODE_LAYOUT");
        bool ok = !(e && atoi(e) == 0);
        for (size_t i = 0; ok && i < names.size(); i++) {
            const std::string& nm = names[i];
            if (c->w.count("fp8dl." + nm)) continue;
            const DevTensor* q8 = WT(c, "fp8." + nm);
            if (!q8 || q8->shape.size() != 2) { ok = false; break; }
            DevTensor t;
            t.dtype = CR_U8; t.shape = {(q8->shape[0] + 15) / 16 * 16, q8->shape[1]}; t.bytes = decode_swizzled8_bytes((int)q8->shape[0], (int)q8->shape[1]);
            if (hipMalloc(&t.ptr, t.bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            if (decode_swizzle_weight8((const unsigned char*)q8->ptr, q8->shape[1], (int)q8->shape[0], (int)q8->shape[1], (unsigned char*)t.ptr, st) != CR_OK) { hipFree(t.ptr); ok = false; break; }
            c->w["fp8dl." + nm] = t;
        }
        if (!ok)
            for (auto it = c->w.begin(); it != c->w.end();) {
                if (it->first.rfind("fp8dl.", 0) == 0) { hipFree(it->second.ptr); it = c->w.erase(it); } else ++it;
            }
    }
    CR_HIP(hipGetLastError());
    CR_HIP(hipStreamSynchronize(st));
    c->fp8_decode = true;
    cr_bump_gen(c);
    return CR_OK;
}

int cr_embed_splice(cr_ctx* c, const int64_t* ids, int S, const void* vit, int n_vit, int64_t img_id, const void* ref,
                    int n_ref, int64_t ref_id, void* out, void* stream) {
    if (!c || !ids || !out || S <= 0) return cr_fail(CR_ERR_ARG, "cr_embed_splice: bad argument");
    if ((size_t)S * 4 + 64 > c->scratch_bytes) return cr_fail(CR_ERR_ARG, "cr_embed_splice: S too large");
    CR_HIP(hipSetDevice(c->device));
    const bf16* table = W(c, "language_model.model.tok_embeddings.weight");
    if (!table) return CR_ERR_STATE;
    hipStream_t st = (hipStream_t)stream;
    int32_t* counts = (int32_t*)c->scratch;
    int32_t* src_row = counts + 16;
    hipLaunchKernelGGL(splice_index_kernel, dim3(1), dim3(1024), 0, st, ids, S, img_id, ref_id, vit ? 1 : 0, ref ? 1 : 0, src_row, counts);
    hipLaunchKernelGGL(splice_rows_kernel, dim3(S), dim3(256), 0, st, table, (const bf16*)vit, (const bf16*)ref, src_row, n_vit, n_ref, (bf16*)out);
    CR_HIP(hipGetLastError());
    // `input_embeds[selected] = vit_embeds` raises on a count mismatch in the reference; check it the same way
    int32_t hc[2];
    CR_HIP(hipMemcpyAsync(hc, counts, 8, hipMemcpyDeviceToHost, st));
    CR_HIP(hipStreamSynchronize(st));
    if (vit && hc[0] != n_vit) return cr_fail(CR_ERR_ARG, "cr_embed_splice: %d <IMG_CONTEXT> ids but %d visual rows", hc[0], n_vit);
    if (vit && hc[0] == 0) return cr_fail(CR_ERR_ARG, "cr_embed_splice: no <IMG_CONTEXT> id in the prompt");
    if (vit && ref && hc[1] != n_ref) return cr_fail(CR_ERR_ARG, "cr_embed_splice: %d pseudo-token ids but %d rows", hc[1], n_ref);
    if (vit && ref && hc[1] == 0) return cr_fail(CR_ERR_ARG, "cr_embed_splice: no pseudo-token id in the prompt");
    return CR_OK;
}

int cr_kv_alloc(cr_ctx* c, int n_seqs, int max_tokens, cr_kv** out) {
    if (!c || !out || n_seqs <= 0 || max_tokens <= 0) return cr_fail(CR_ERR_ARG, "cr_kv_alloc: bad argument");
    if (max_tokens > c->d.max_pos) return cr_fail(CR_ERR_ARG, "cr_kv_alloc: max_tokens %d exceeds the RoPE table (%d rows)", max_tokens, c->d.max_pos);
    CR_HIP(hipSetDevice(c->device));
    cr_kv* kv = new cr_kv();
    kv->ctx = c; kv->n_seqs = n_seqs; kv->max_tokens = max_tokens; kv->gen_cap = 4096; kv->layers = c->d.llm_layers;
    const size_t bytes = (size_t)kv->layers * n_seqs * NKV * max_tokens * HD * 2;
    if (hipMalloc((void**)&kv->k, bytes) != hipSuccess || hipMalloc((void**)&kv->v, bytes) != hipSuccess ||
        hipMalloc((void**)&kv->d_len, n_seqs * 4) != hipSuccess || hipMalloc((void**)&kv->d_ngen, n_seqs * 4) != hipSuccess ||
        hipMalloc((void**)&kv->d_seqs, n_seqs * 4) != hipSuccess ||
        hipMalloc((void**)&kv->d_gen, (size_t)n_seqs * kv->gen_cap * 8) != hipSuccess) {
        cr_kv_free(kv);
        return cr_fail(CR_ERR_NOMEM, "cr_kv_alloc: %zu bytes per K/V", bytes);
    }
    CR_HIP(hipMemset(kv->d_len, 0, n_seqs * 4));
    CR_HIP(hipMemset(kv->d_ngen, 0, n_seqs * 4));
    kv->len.assign(n_seqs, 0); kv->ngen.assign(n_seqs, 0);
    *out = kv;
    return CR_OK;
}

int cr_kv_free(cr_kv* kv) {
    if (!kv) return CR_OK;
    hipDeviceSynchronize();
    if (kv->k) hipFree(kv->k);
    if (kv->v) hipFree(kv->v);
    if (kv->d_len) hipFree(kv->d_len);
    if (kv->d_ngen) hipFree(kv->d_ngen);
    if (kv->d_seqs) hipFree(kv->d_seqs);
    if (kv->d_gen) hipFree(kv->d_gen);
    for (auto& e : kv->graphs) if (e.exec) hipGraphExecDestroy(e.exec);
    delete kv;
    return CR_OK;
}

int cr_kv_length(const cr_kv* kv, int seq) { return (kv && seq >= 0 && seq < kv->n_seqs) ? kv->len[seq] : -1; }

int cr_kv_reset(cr_kv* kv, int seq, void* stream) {
    if (!kv || seq >= kv->n_seqs) return cr_fail(CR_ERR_ARG, "cr_kv_reset: bad sequence");
    CR_HIP(hipSetDevice(kv->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int s0 = seq < 0 ? 0 : seq, n = seq < 0 ? kv->n_seqs : 1;
    CR_HIP(hipMemsetAsync(kv->d_len + s0, 0, (size_t)n * 4, st));
    CR_HIP(hipMemsetAsync(kv->d_ngen + s0, 0, (size_t)n * 4, st));
    for (int s = s0; s < s0 + n; s++) { kv->len[s] = 0; kv->ngen[s] = 0; }
    return CR_OK;
}

int cr_kv_read(cr_kv* kv, int layer, int seq, int pos, int which, void* out, void* stream) {
    if (!kv || !out || layer < 0 || layer >= kv->layers || seq < 0 || seq >= kv->n_seqs || pos < 0 || pos >= kv->max_tokens || (which != 0 && which != 1))
        return cr_fail(CR_ERR_ARG, "cr_kv_read: bad argument");
    CR_HIP(hipSetDevice(kv->ctx->device));
    const bf16* base = (which ? kv->v : kv->k) + (((int64_t)layer * kv->n_seqs + seq) * NKV * kv->max_tokens + pos) * HD;
    CR_HIP(hipMemcpy2DAsync(out, HD * 2, base, (size_t)kv->max_tokens * HD * 2, HD * 2, NKV, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CR_OK;
}

int cr_kv_generated(cr_kv* kv, int seq, int64_t* out_host, int max, void* stream) {
    if (!kv || !out_host || seq < 0 || seq >= kv->n_seqs) return cr_fail(CR_ERR_ARG, "cr_kv_generated: bad argument");
    const int n = kv->ngen[seq] < max ? kv->ngen[seq] : max;
    if (n > 0) CR_HIP(hipMemcpyAsync(out_host, kv->d_gen + (size_t)seq * kv->gen_cap, (size_t)n * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return n;
}

int cr_llm_prefill_batch(cr_ctx* c, cr_kv* kv, const int32_t* seqs, int n, const void* embeds, const int32_t* lens, float penalty,
                         float* last_logits, void* stream) {
    if (!c || !kv || !seqs || !lens || !embeds || n <= 0 || n > kv->n_seqs) return cr_fail(CR_ERR_ARG, "cr_llm_prefill_batch: bad argument");
    if (!c->finalized) return cr_fail(CR_ERR_STATE, "cr_llm_prefill_batch: call cr_finalize first");
    CR_TRY(ctx_share_ok(c, "cr_llm_prefill_batch"));
    std::vector<Segment> segs;
    int M = 0;
    for (int i = 0; i < n; i++) {
        const int s = seqs[i];
        if (s < 0 || s >= kv->n_seqs || lens[i] <= 0) return cr_fail(CR_ERR_ARG, "cr_llm_prefill_batch: bad sequence %d / length %d", s, lens[i]);
        for (int j = 0; j < i; j++) if (seqs[j] == s) return cr_fail(CR_ERR_ARG, "cr_llm_prefill_batch: sequence %d listed twice", s);
        if (kv->len[s] + lens[i] > kv->max_tokens)
            return cr_fail(CR_ERR_ARG, "cr_llm_prefill: %d + %d tokens exceed the cache (%d)", kv->len[s], lens[i], kv->max_tokens);
        segs.push_back({s, M, lens[i], kv->len[s]});
        M += lens[i];
    }
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int V = c->d.vocab;
    const size_t lw = layers_ws(c, M, 0, n);
    CR_TRY(ws_ensure(c, lw + (size_t)M * D * 2 + (size_t)n * D * 2 + (size_t)n * V * 4 + (size_t)M * 8 + (size_t)n * 36 + 8192));
    bf16* x = (bf16*)(c->ws + ((lw + 255) & ~(size_t)255));
    bf16* hl = x + (size_t)M * D;
    float* lg = (float*)(((uintptr_t)(hl + (size_t)n * D) + 255) & ~(uintptr_t)255);
    int32_t* d_row_seq = (int32_t*)(((uintptr_t)(lg + (size_t)n * V) + 255) & ~(uintptr_t)255);
    int32_t* d_row_pos = d_row_seq + M;
    int32_t* d_seg = d_row_pos + M;                 // 4 per page: first row, rows, first position, cache slot (the attention launch's segments)
    int32_t* d_seg_last = d_seg + 4 * n;            // the final layer's segments: the rows that share a 32-row wave with the page's last row
    int32_t* d_last = d_seg_last + 4 * n;           // and that last row
    {
        std::vector<int32_t> h(2 * (size_t)M + 9 * segs.size());
        for (const Segment& sg : segs)
            for (int r = 0; r < sg.S; r++) { h[sg.row0 + r] = sg.seq; h[(size_t)M + sg.row0 + r] = sg.pos0 + r; }
        for (size_t i = 0; i < segs.size(); i++) {
            int32_t* e = h.data() + 2 * (size_t)M + 4 * i;
            e[0] = segs[i].row0; e[1] = segs[i].S; e[2] = segs[i].pos0; e[3] = segs[i].seq;
            // the attention kernel's query blocks are 128 rows of a segment, its waves 32: rows [32 k, 32 k + 32) of a page meet in one wave
            const int w0 = (segs[i].S - 1) / 32 * 32;
            int32_t* f = h.data() + 2 * (size_t)M + 4 * segs.size() + 4 * i;
            f[0] = segs[i].row0 + w0; f[1] = segs[i].S - w0; f[2] = segs[i].pos0 + w0; f[3] = segs[i].seq;
            h[2 * (size_t)M + 8 * segs.size() + i] = segs[i].row0 + segs[i].S - 1;
        }
        CR_HIP(hipMemcpyAsync(d_row_seq, h.data(), h.size() * 4, hipMemcpyHostToDevice, st));   // pageable source: staged before return
    }
    CR_HIP(hipMemcpyAsync(kv->d_seqs, seqs, (size_t)n * 4, hipMemcpyHostToDevice, st));
    kv->seqs_on_device.assign(seqs, seqs + n);
    CR_HIP(hipMemcpyAsync(x, embeds, (size_t)M * D * 2, hipMemcpyDeviceToDevice, st));
    CR_TRY(run_layers(c, kv, x, M, false, segs, d_row_seq, d_row_pos, nullptr, 0, st, d_seg, d_last, d_seg_last));
    const bf16 *nw = W(c, "language_model.model.norm.weight"), *ow = W(c, "language_model.output.weight");
    if (!nw || !ow) return CR_ERR_STATE;
    // only each page's last row feeds the LM head (the reference computes all S rows and reads the last, :1081 + _sample)
    for (int i = 0; i < n; i++)
        CR_TRY(rms(x + (size_t)(segs[i].row0 + segs[i].S - 1) * D, D, hl + (size_t)i * D, nw, 1, c->d.rms_eps, st));
    CR_TRY(gemm(c, EPI_F32, hl, D, ow, D, lg, V, nullptr, 0, n, V, D, st));
    if (last_logits) CR_HIP(hipMemcpyAsync(last_logits, lg, (size_t)n * V * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(pick_kernel, dim3(n), dim3(1024), 0, st, lg, (int64_t)V, V, penalty, 0, kv->d_seqs, kv->d_gen, kv->d_ngen,
                       kv->d_len, kv->gen_cap, 0);
    CR_HIP(hipGetLastError());
    for (const Segment& sg : segs) {
        kv->len[sg.seq] += sg.S;
        if (kv->ngen[sg.seq] < kv->gen_cap) kv->ngen[sg.seq] += 1;
    }
    CR_HIP(hipMemcpyAsync(kv->d_len, kv->len.data(), (size_t)kv->n_seqs * 4, hipMemcpyHostToDevice, st));
    return CR_OK;
}

int cr_llm_hidden_probe(cr_ctx* c, void* dst, int row0, int rows) {
    if (!c || (dst && (row0 < 0 || rows <= 0))) return cr_fail(CR_ERR_ARG, "cr_llm_hidden_probe: bad argument");
    c->probe_dst = (bf16*)dst; c->probe_row0 = row0; c->probe_rows = dst ? rows : 0;
    return CR_OK;
}

int cr_llm_prefill(cr_ctx* c, cr_kv* kv, int seq, const void* embeds, int S, float penalty, float* last_logits, void* stream) {
    if (!kv || S <= 0) return cr_fail(CR_ERR_ARG, "cr_llm_prefill: bad argument");
    const int32_t s = seq, l = S;
    return cr_llm_prefill_batch(c, kv, &s, 1, embeds, &l, penalty, last_logits, stream);
}

int cr_llm_decode(cr_ctx* c, cr_kv* kv, const int32_t* seqs, int n, const int64_t* force_tokens, float penalty, float* logits,
                  void* stream) {
    if (!c || !kv || !seqs || n <= 0 || n > kv->n_seqs) return cr_fail(CR_ERR_ARG, "cr_llm_decode: bad argument");
    if (!c->finalized) return cr_fail(CR_ERR_STATE, "cr_llm_decode: call cr_finalize first");
    CR_TRY(ctx_share_ok(c, "cr_llm_decode"));
    for (int i = 0; i < n; i++) {
        const int s = seqs[i];
        if (s < 0 || s >= kv->n_seqs) return cr_fail(CR_ERR_ARG, "cr_llm_decode: sequence %d out of range", s);
        if (kv->len[s] + 1 > kv->max_tokens) return cr_fail(CR_ERR_ARG, "cr_llm_decode: sequence %d cache full", s);
        if (!force_tokens && kv->ngen[s] == 0) return cr_fail(CR_ERR_STATE, "cr_llm_decode: sequence %d has no generated id to feed", s);
        for (int j = 0; j < i; j++) if (seqs[j] == s) return cr_fail(CR_ERR_ARG, "cr_llm_decode: sequence %d listed twice", s);
    }
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int V = c->d.vocab;
    int max_len = 0;
    for (int i = 0; i < n; i++) max_len = kv->len[seqs[i]] > max_len ? kv->len[seqs[i]] : max_len;
    const int nsplit = ((max_len + 1 + 63) / 64 + ATTN_SPLIT_TILES - 1) / ATTN_SPLIT_TILES;
    const size_t lw = layers_ws(c, n, nsplit);
    CR_TRY(ws_ensure(c, lw + (size_t)n * D * 4 + (size_t)n * V * 4 + 4096));
    bf16* x = (bf16*)(c->ws + ((lw + 255) & ~(size_t)255));
    bf16* hl = x + (size_t)n * D;
    float* lg = (float*)(((uintptr_t)(hl + (size_t)n * D) + 255) & ~(uintptr_t)255);
    if ((int)kv->seqs_on_device.size() != n || memcmp(kv->seqs_on_device.data(), seqs, (size_t)n * 4) != 0) {      // prefill uses d_seqs too
        CR_HIP(hipMemcpyAsync(kv->d_seqs, seqs, (size_t)n * 4, hipMemcpyHostToDevice, st));
        kv->seqs_on_device.assign(seqs, seqs + n);
    }
    const bf16* table = W(c, "language_model.model.tok_embeddings.weight");
    const bf16 *nw = W(c, "language_model.model.norm.weight"), *ow = W(c, "language_model.output.weight");
    if (!table || !nw || !ow) return CR_ERR_STATE;
    auto enqueue_step = [&](hipStream_t st) -> int {
        hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(256), 0, st, table, force_tokens, kv->d_seqs, kv->d_gen, kv->d_ngen, kv->gen_cap, x);
        CR_TRY(run_layers(c, kv, x, n, true, {}, nullptr, nullptr, kv->d_seqs, nsplit, st));
        const DevTensor *q_out = opt(c, "fp8.language_model.output.weight"), *s_out = opt(c, "fp8s.language_model.output.weight");
        const int ff_ = (int)WT(c, "derived.w13.0")->shape[0] / 2;
        if (c->fused_decode && !c->fp8_decode && !c->no_sliced_decode && decode_fused_supported(n, ff_)) {
            DecodeGemmParams dh{};                                // final RMSNorm folded into the LM head (gemm_decode.hip)
            dh.M = n; dh.eps = c->d.rms_eps; dh.W = ow; dh.ldw = D; dh.N = V; dh.K = D; dh.xres = x; dh.gamma = nw; dh.C = lg; dh.ldc = V;
            if (const DevTensor* dl = opt(c, "declayout.language_model.output.weight")) { dh.W = (const bf16*)dl->ptr; dh.swizzled = 1; }
            CR_TRY(launch_decode_gemm(DEC_HEAD, dh, st));
        } else {
        CR_TRY(rms(x, D, hl, nw, n, c->d.rms_eps, st));
        if (c->fp8_decode && n <= 64 && q_out && s_out) CR_TRY(gemm8(c, EPI_F32, hl, D, q_out, s_out, lg, V, nullptr, 0, n, V, D, st, opt(c, "fp8dl.language_model.output.weight")));
        else CR_TRY(gemm(c, EPI_F32, hl, D, ow, D, lg, V, nullptr, 0, n, V, D, st, false, opt(c, "declayout.language_model.output.weight")));
        }
        if (logits) CR_HIP(hipMemcpyAsync(logits, lg, (size_t)n * V * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(pick_kernel, dim3(n), dim3(1024), 0, st, lg, (int64_t)V, V, penalty, 0, kv->d_seqs, kv->d_gen, kv->d_ngen,
                           kv->d_len, kv->gen_cap, 1);
        return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
    };
    // graph path: the plain generation loop (no forced ids, no logits copy), nothing recording events on these launches
    const bool graphable = c->decode_graph && !force_tokens && !logits && !(c->prof && c->prof_mode != 2);
    bool done = false;
    hipStream_t gs = st;                               // the null stream cannot be captured: use the context's blocking side stream
    if (graphable && !st) {
        if (!c->side && hipStreamCreate(&c->side) != hipSuccess) { (void)hipGetLastError(); c->decode_graph = 0; }
        gs = c->side;
    }
    if (getenv("CR_GRAPH_DEBUG") && !graphable) fprintf(stderr, "[cr] decode not graphable: flag %d force %d logits %d prof %d/%d\n", c->decode_graph, force_tokens != nullptr, logits != nullptr, (int)c->prof, c->prof_mode);
    if (graphable && c->decode_graph && gs) {
        cr_kv::DecodeGraph* g = nullptr;
        for (auto& e : kv->graphs)
            if (e.n == n && e.nsplit == nsplit && e.penalty == penalty && e.ws == c->ws && e.weight_gen == c->weight_gen) { g = &e; break; }
        if (!g) {
            if (kv->graphs.size() >= 24) {                       // stale keys (grown workspace, reloaded weights, old split counts)
                for (auto& e : kv->graphs) if (e.exec) hipGraphExecDestroy(e.exec);
                kv->graphs.clear();
            }
            kv->graphs.push_back({n, nsplit, penalty, c->ws, c->weight_gen, nullptr, 0});
            g = &kv->graphs.back();
        }
        if (g->exec) {
            if (hipGraphLaunch(g->exec, gs) == hipSuccess) done = true;
            else { (void)hipGetLastError(); c->decode_graph = 0; }
        } else if (g->warm >= 1) {
            // second step with this key: one-time attribute set-up and workspace growth happened in the eager first one
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = enqueue_step(gs);
                const hipError_t ee = hipStreamEndCapture(gs, &graph);
                if (rc == CR_OK && ee == hipSuccess && graph && hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess &&
                    hipGraphLaunch(g->exec, gs) == hipSuccess) {
                    done = true;
                    if (getenv("CR_GRAPH_DEBUG")) fprintf(stderr, "[cr] decode graph captured: rows %d, splits %d\n", n, nsplit);
                } else {
                    if (getenv("CR_GRAPH_DEBUG")) fprintf(stderr, "[cr] decode graph capture FAILED (rc %d, end %d): %s\n", rc, (int)ee, hipGetErrorString(hipGetLastError()));
                    (void)hipGetLastError();
                    if (g->exec) { hipGraphExecDestroy(g->exec); g->exec = nullptr; }
                    c->decode_graph = 0;                          // fall back to plain launches for good
                }
                if (graph) hipGraphDestroy(graph);
            } else {
                if (getenv("CR_GRAPH_DEBUG")) fprintf(stderr, "[cr] hipStreamBeginCapture failed: %s\n", hipGetErrorString(hipGetLastError()));
                (void)hipGetLastError();
                c->decode_graph = 0;
            }
        } else {
            g->warm++;
        }
    }
    if (!done) CR_TRY(enqueue_step(st));
    for (int i = 0; i < n; i++) {
        kv->len[seqs[i]] += 1;
        if (kv->ngen[seqs[i]] < kv->gen_cap) kv->ngen[seqs[i]] += 1;
    }
    return CR_OK;
}

}  // extern "C"
