// Vision stage: InternViT-300M encoder + pixel-shuffle/mlp1 projector.
//   reference: InternVL/modeling_intern_vit.py:138-437, InternVL/modeling_internvl_chat.py:283-319
//
// Per tile chunk (<= VIT_CHUNK tiles, M = tiles * 1025 rows) and per layer, 7 launches:
//   LN1 -> GEMM qkv(+bias) -> flash attention -> GEMM proj(+bias, *ls1, +x, in place)
//   LN2 -> GEMM fc1(+bias, GELU) -> GEMM fc2(+bias, *ls2, +x, in place)
// The residual stream x lives in the caller's output buffer; nothing is copied.
#include <stdlib.h>

#include "attention.hpp"
#include "ctx.hpp"
#include "misc.hpp"
#include "norm.hpp"

static constexpr int VIT_CHUNK = 255;  // 255*1025 rows = 1021 row-tiles of 256: x4 column tiles = 15.95 rounds of 256 CUs; per-launch fixed costs (cold
                                       // start, tail) spread over 4x the work of round 1's 63-tile chunks: 1.3 % on 510 tiles; operands stay < 4 GiB
// Chunks of 16k-1 tiles (127, ..., 31, 15) put ceil(rows/256) * {4,12,16} GEMM tiles just under a whole number of
// 256-CU rounds.  Up to 64 tiles go as ONE chunk: rows = 1024 T + T, so the T rows past the last multiple of 256 are exactly
// what the dispatcher hands to the one-wave tail kernel (gemm.hip: launch_gemm_tail, <= 64 rows) -- 32 tiles (BASELINE config 2)
// used to run as 31 + 1, and the 1-tile chunk's 24 x 7 launches of almost nothing cost more than the 32nd tile's work.
static int next_chunk(int remaining) {
    static const int cap = [] { const char* e = getenv("CR_VIT_CHUNK"); const int v = e ? atoi(e) : 0; return v > 0 && v < VIT_CHUNK ? v : VIT_CHUNK; }();   // tuning aid
    if (remaining <= 64 && remaining <= cap) return remaining;
    for (int c : {255, 191, 127, 111, 95, 79, 63, 47, 31, 15}) if (c <= cap && remaining >= c) return c;
    return remaining < cap ? remaining : cap;
}
static constexpr int C1 = 1024, C3 = 3072, FF = 4096, TOK = 1025, KPAD = 640;

int vit_finalize(cr_ctx* c, hipStream_t st) {
    const DevTensor* pw = WT(c, "vision_model.embeddings.patch_embedding.weight");
    if (!pw) return CR_ERR_STATE;
    if (pw->numel() != (int64_t)C1 * 588) return cr_fail(CR_ERR_ARG, "patch_embedding.weight must be [1024,3,14,14]");
    DevTensor t;
    t.dtype = CR_BF16; t.shape = {C1, KPAD}; t.bytes = (size_t)C1 * KPAD * 2;
    auto it = c->w.find("derived.patch_w");
    if (it != c->w.end()) t.ptr = it->second.ptr;
    else CR_HIP(hipMalloc(&t.ptr, t.bytes));
    CR_HIP(hipMemsetAsync(t.ptr, 0, t.bytes, st));
    CR_HIP(hipMemcpy2DAsync(t.ptr, KPAD * 2, pw->ptr, 588 * 2, 588 * 2, C1, hipMemcpyDeviceToDevice, st));
    c->w["derived.patch_w"] = t;
    return CR_OK;
}

static const DevTensor* WTopt(cr_ctx* c, const std::string& name) {
    auto it = c->w.find(name);
    return it == c->w.end() ? nullptr : &it->second;
}

static int gemm(cr_ctx* c, int epi, const bf16* A, int64_t lda, const bf16* Wt, int64_t ldw, void* C, int64_t ldc, const bf16* bias,
                const bf16* scale, const bf16* res, int64_t ldr, int M, int N, int K, int group, hipStream_t st) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = Wt; p.ldw = ldw; p.C = C; p.ldc = ldc; p.bias = bias; p.scale = scale; p.res = res; p.ldr = ldr;
    p.M = M; p.N = N; p.K = K; p.group = group;
    return ctx_gemm(c, epi, p, st);
}

static int vit_chunk(cr_ctx* c, const bf16* px, int T, bf16* x, hipStream_t st) {
    const int M = T * TOK;
    Arena ar(c->ws);
    bf16* col = ar.take<bf16>((size_t)T * 1024 * KPAD);
    bf16* h = ar.take<bf16>((size_t)M * C1);       // LN output, later attention output
    bf16* qkv = ar.take<bf16>((size_t)M * C3);
    bf16* f = ar.take<bf16>((size_t)M * FF);
    float* hs = ar.take<float>((size_t)M);          // fp8 matrix-core path: one scale per LayerNorm row
    float* fs = ar.take<float>((size_t)M);          // ... and per row of fc1's e4m3 output
    float* apart = ar.take<float>(vit_attn_ws_floats(T, 16, TOK));     // attention_vit.hip: the CLS query's partials

    const bf16* patch_w = W(c, "derived.patch_w");
    const bf16* patch_b = W(c, "vision_model.embeddings.patch_embedding.bias");
    const bf16* cls = W(c, "vision_model.embeddings.class_embedding");
    const bf16* pos = W(c, "vision_model.embeddings.position_embedding");
    if (!patch_w || !patch_b || !cls || !pos) return CR_ERR_STATE;

    // embeddings (modeling_intern_vit.py:167-179); bicubic pos-emb resize is the identity at 448x448
    CR_TRY(launch_im2col14(px, col, T, st));
    CR_TRY(gemm(c, EPI_PATCH, col, KPAD, patch_w, KPAD, x, C1, patch_b, nullptr, pos, C1, T * 1024, C1, KPAD, 1024, st));
    CR_TRY(launch_cls_rows(cls, pos, x, T, C1, TOK, st));

    for (int l = 0; l < c->d.vit_layers; l++) {
        const std::string p = "vision_model.encoder.layers." + std::to_string(l) + ".";
        const bf16 *n1w = W(c, p + "norm1.weight"), *n1b = W(c, p + "norm1.bias");
        const bf16 *n2w = W(c, p + "norm2.weight"), *n2b = W(c, p + "norm2.bias");
        const bf16 *qkvw = W(c, p + "attn.qkv.weight"), *qkvb = W(c, p + "attn.qkv.bias");
        const bf16 *pw = W(c, p + "attn.proj.weight"), *pb = W(c, p + "attn.proj.bias");
        const bf16 *f1w = W(c, p + "mlp.fc1.weight"), *f1b = W(c, p + "mlp.fc1.bias");
        const bf16 *f2w = W(c, p + "mlp.fc2.weight"), *f2b = W(c, p + "mlp.fc2.bias");
        const bf16 *ls1 = W(c, p + "ls1"), *ls2 = W(c, p + "ls2");
        if (!n1w || !n1b || !n2w || !n2b || !qkvw || !qkvb || !pw || !pb || !f1w || !f1b || !f2w || !f2b || !ls1 || !ls2)
            return CR_ERR_STATE;

        // cr_enable_fp8_mfma: the two LayerNorms emit e4m3 rows + scales (into h) and QKV / fc1 multiply e4m3 x e4m3
        const DevTensor *q_qkv = WTopt(c, "fp8." + p + "attn.qkv.weight"), *s_qkv = WTopt(c, "fp8s." + p + "attn.qkv.weight");
        const DevTensor *q_f1 = WTopt(c, "fp8." + p + "mlp.fc1.weight"), *s_f1 = WTopt(c, "fp8s." + p + "mlp.fc1.weight");
        const DevTensor *q_f2 = WTopt(c, "fp8." + p + "mlp.fc2.weight"), *s_f2 = WTopt(c, "fp8s." + p + "mlp.fc2.weight");
        const DevTensor* bnd = WTopt(c, "fp8b." + p + "mlp.fc1.weight");
        const bool m8 = c->fp8_mfma && q_qkv && s_qkv && q_f1 && s_f1;
        const bool m8all = m8 && c->fp8_mfma >= 2 && q_f2 && s_f2 && bnd;      // level 2: fc1 hands fc2 e4m3 rows straight from its epilogue

        NormParams np{};
        np.in = x; np.ld_in = C1; np.out = h; np.ld_out = C1; np.rows = M; np.eps = c->d.vit_ln_eps;
        np.gamma = n1w; np.beta = n1b;
        if (m8) { np.out8 = (unsigned char*)h; np.out8_scale = hs; }
        CR_TRY(launch_layernorm(np, C1, 0, st));
        if (m8) CR_TRY(ctx_gemm_f8(c, EPI_STORE, h, hs, q_qkv, s_qkv, qkv, C3, qkvb, M, C3, C1, st));
        else CR_TRY(gemm(c, EPI_STORE, h, C1, qkvw, C1, qkv, C3, qkvb, nullptr, nullptr, 0, M, C3, C1, 0, st));

        AttnParams ap{};
        ap.Q = qkv; ap.K = qkv + C1; ap.V = qkv + 2 * C1; ap.O = h;
        ap.q_bs = ap.k_bs = ap.v_bs = (int64_t)TOK * C3; ap.q_rs = ap.k_rs = ap.v_rs = C3; ap.q_hs = ap.k_hs = ap.v_hs = 64;
        ap.o_bs = (int64_t)TOK * C1; ap.o_rs = C1; ap.o_hs = 64;
        ap.B = T; ap.H = 16; ap.Sq = TOK; ap.Sk = TOK; ap.kv_group = 1; ap.q_pos0 = 0;
        ap.q_prescale = 0.125f; ap.s_div = 1.0f;
        ap.part_ml = apart;
        if (launch_flash_attn(ap, 64, false, st) != CR_OK) return cr_fail(CR_ERR_HIP, "vit attention launch failed");
        CR_TRY(gemm(c, EPI_LS_RES, h, C1, pw, C1, x, C1, pb, ls1, x, C1, M, C1, C1, 0, st));

        np.gamma = n2w; np.beta = n2b;
        if (m8all) { np.next_scale = fs; np.next_bound = (const float*)bnd->ptr; }
        CR_TRY(launch_layernorm(np, C1, 0, st));
        if (m8 && !m8all) {
            // level 1: fc1 (its input is LayerNorm 2's e4m3 row) multiplies e4m3 x e4m3 and writes bf16; fc2 stays bf16
            CR_TRY(ctx_gemm_f8(c, EPI_GELU, h, hs, q_f1, s_f1, f, FF, f1b, M, FF, C1, st));
            CR_TRY(gemm(c, EPI_LS_RES, f, FF, f2w, FF, x, C1, f2b, ls2, x, C1, M, C1, FF, 0, st));
        } else if (m8all) {
            // fc1 in e4m3, its GELU output written as e4m3 rows (scale = the bound LayerNorm 2 derived), fc2 in e4m3 on those rows
            GemmParams g1{};
            g1.A = h; g1.lda = C1; g1.W = (const bf16*)q_f1->ptr; g1.ldw = C1; g1.C = f; g1.ldc = FF; g1.bias = f1b; g1.M = M; g1.N = FF; g1.K = C1;
            g1.w8 = 1; g1.wscale = (const float*)s_f1->ptr; g1.a8 = 1; g1.ascale = hs; g1.c8scale = fs;
            CR_TRY(ctx_gemm(c, EPI_GELU_Q8, g1, st));
            GemmParams g2{};
            g2.A = f; g2.lda = FF; g2.W = (const bf16*)q_f2->ptr; g2.ldw = FF; g2.C = x; g2.ldc = C1; g2.bias = f2b; g2.scale = ls2; g2.res = x; g2.ldr = C1;
            g2.M = M; g2.N = C1; g2.K = FF; g2.w8 = 1; g2.wscale = (const float*)s_f2->ptr; g2.a8 = 1; g2.ascale = fs;
            CR_TRY(ctx_gemm(c, EPI_LS_RES, g2, st));
        } else {
            CR_TRY(gemm(c, EPI_GELU, h, C1, f1w, C1, f, FF, f1b, nullptr, nullptr, 0, M, FF, C1, 0, st));
            CR_TRY(gemm(c, EPI_LS_RES, f, FF, f2w, FF, x, C1, f2b, ls2, x, C1, M, C1, FF, 0, st));
        }
    }
    return CR_OK;
}

static size_t vit_ws_bytes(int T) {
    const size_t M = (size_t)T * TOK;
    return ((size_t)T * 1024 * KPAD + M * C1 + M * C3 + M * FF) * 2 + M * 8 + vit_attn_ws_floats(T, 16, TOK) * 4 + 8192;
}

static size_t project_ws_bytes(int T) { return (size_t)T * 256 * 4096 * 4 + (size_t)T * 256 * 4 + 8192; }   // a, b, row scales

static int project_chunk(cr_ctx* c, const bf16* vit_out, int T, bf16* out, hipStream_t st) {
    const int M = T * 256;
    Arena ar(c->ws);
    bf16* a = ar.take<bf16>((size_t)M * 4096);
    bf16* b = ar.take<bf16>((size_t)M * 4096);
    const bf16 *lw = W(c, "mlp1.0.weight"), *lb = W(c, "mlp1.0.bias");
    const bf16 *w1 = W(c, "mlp1.1.weight"), *b1 = W(c, "mlp1.1.bias");
    const bf16 *w3 = W(c, "mlp1.3.weight"), *b3 = W(c, "mlp1.3.bias");
    if (!lw || !lb || !w1 || !b1 || !w3 || !b3) return CR_ERR_STATE;
    const DevTensor *q_w1 = WTopt(c, "fp8.mlp1.1.weight"), *s_w1 = WTopt(c, "fp8s.mlp1.1.weight");
    const bool m8 = c->fp8_mfma && q_w1 && s_w1;
    float* as = ar.take<float>((size_t)M);
    NormParams np{};
    np.in = vit_out; np.out = a; np.ld_out = 4096; np.rows = M; np.eps = 1e-5f; np.gamma = lw; np.beta = lb;
    if (m8) { np.out8 = (unsigned char*)a; np.out8_scale = as; }
    CR_TRY(launch_layernorm(np, 4096, 1, st));     // drop CLS + pixel_shuffle folded into the load
    if (m8) CR_TRY(ctx_gemm_f8(c, EPI_GELU, a, as, q_w1, s_w1, b, 4096, b1, M, 4096, 4096, st));
    else CR_TRY(gemm(c, EPI_GELU, a, 4096, w1, 4096, b, 4096, b1, nullptr, nullptr, 0, M, 4096, 4096, 0, st));
    CR_TRY(gemm(c, EPI_STORE, b, 4096, w3, 4096, out, 4096, b3, nullptr, nullptr, 0, M, 4096, 4096, 0, st));
    return CR_OK;
}

extern "C" {

int cr_vit_forward(cr_ctx* c, const void* pixels, int T, void* out, void* stream) {
    if (!c || !pixels || !out || T <= 0) return cr_fail(CR_ERR_ARG, "cr_vit_forward: bad argument");
    if (!c->finalized) return cr_fail(CR_ERR_STATE, "cr_vit_forward: call cr_finalize after loading weights");
    CR_TRY(ctx_share_ok(c, "cr_vit_forward"));
    CR_HIP(hipSetDevice(c->device));
    CR_TRY(ws_ensure(c, vit_ws_bytes(T < VIT_CHUNK ? T : VIT_CHUNK)));
    for (int t0 = 0; t0 < T;) {
        const int tc = next_chunk(T - t0);
        CR_TRY(vit_chunk(c, (const bf16*)pixels + (size_t)t0 * 3 * 448 * 448, tc, (bf16*)out + (size_t)t0 * TOK * C1,
                         (hipStream_t)stream));
        t0 += tc;
    }
    return CR_OK;
}

int cr_project(cr_ctx* c, const void* vit_out, int T, void* out, void* stream) {
    if (!c || !vit_out || !out || T <= 0) return cr_fail(CR_ERR_ARG, "cr_project: bad argument");
    CR_HIP(hipSetDevice(c->device));
    const int CH = 256;
    CR_TRY(ws_ensure(c, project_ws_bytes(T < CH ? T : CH)));
    for (int t0 = 0; t0 < T; t0 += CH) {
        const int tc = (T - t0) < CH ? (T - t0) : CH;
        CR_TRY(project_chunk(c, (const bf16*)vit_out + (size_t)t0 * TOK * C1, tc, (bf16*)out + (size_t)t0 * 256 * 4096,
                             (hipStream_t)stream));
    }
    return CR_OK;
}

int cr_extract_feature(cr_ctx* c, const void* pixels, int T, void* out, void* stream) {
    if (!c || !pixels || !out || T <= 0) return cr_fail(CR_ERR_ARG, "cr_extract_feature: bad argument");
    if (!c->finalized) return cr_fail(CR_ERR_STATE, "cr_extract_feature: call cr_finalize after loading weights");
    CR_TRY(ctx_share_ok(c, "cr_extract_feature"));
    CR_HIP(hipSetDevice(c->device));
    // ViT output of a chunk sits at the top of the workspace, below it the per-chunk scratch of both stages.
    for (int t0 = 0; t0 < T;) {
        const int tc = next_chunk(T - t0);
        const size_t inner = vit_ws_bytes(tc) > project_ws_bytes(tc) ? vit_ws_bytes(tc) : project_ws_bytes(tc);
        const size_t xbytes = (size_t)tc * TOK * C1 * 2;
        CR_TRY(ws_ensure(c, inner + 256 + xbytes));
        bf16* x = (bf16*)(c->ws + ((inner + 255) & ~(size_t)255));
        CR_TRY(vit_chunk(c, (const bf16*)pixels + (size_t)t0 * 3 * 448 * 448, tc, x, (hipStream_t)stream));
        CR_TRY(project_chunk(c, x, tc, (bf16*)out + (size_t)t0 * 256 * 4096, (hipStream_t)stream));
        t0 += tc;
    }
    return CR_OK;
}

}  // extern "C"
