// C ABI of libcallireader_hip.so: lifetime, weights, single-operator entry points.
// Stage orchestration lives in vision.hip / calli.hip / llm.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "attention.hpp"
#include <stdlib.h>

#include "ctx.hpp"
#include "decode.hpp"
#include "misc.hpp"
#include "norm.hpp"

static thread_local char g_err[512] = "";

void cr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int cr_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const DevTensor* WT(cr_ctx* c, const std::string& name) {
    auto it = c->w.find(name);
    if (it == c->w.end()) { cr_set_error("weight '%s' was never loaded", name.c_str()); return nullptr; }
    return &it->second;
}
const bf16* W(cr_ctx* c, const std::string& name) {
    const DevTensor* t = WT(c, name);
    return t ? (const bf16*)t->ptr : nullptr;
}

int ws_ensure(cr_ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) return CR_OK;
    // growth happens only the first time a shape is seen; never inside a timed steady state
    CR_HIP(hipDeviceSynchronize());
    if (c->ws) CR_HIP(hipFree(c->ws));
    c->ws = nullptr; c->ws_bytes = 0;
    size_t want = bytes + (bytes >> 3);
    if (hipMalloc((void**)&c->ws, want) != hipSuccess) return cr_fail(CR_ERR_NOMEM, "workspace of %zu bytes", want);
    c->ws_bytes = want;
    return CR_OK;
}

// Measurement (cr_profile): one event pair per GEMM launch on the launch stream.  Records are retired into per-class
// accumulators as soon as their second event has completed (events of one stream complete in order, so the queue is
// drained from the front with a non-blocking hipEventQuery); nothing is ever dropped.  Only when more than
// PROF_HARD records are still in flight (the host running that far ahead of the GPU) does the oldest one get waited for.
static constexpr size_t PROF_SOFT = 256, PROF_HARD = 16384;

static void prof_retire(cr_ctx* c, bool block_all) {
    while (!c->prof_recs.empty()) {
        cr_ctx::ProfRec& r = c->prof_recs.front();
        hipError_t q = hipEventQuery(r.b);
        if (q == hipErrorNotReady) {
            if (!block_all && c->prof_recs.size() <= PROF_HARD) break;
            q = hipEventSynchronize(r.b);
        }
        float ms = 0.f;
        if (q == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            double* o = c->prof_acc[r.big ? 0 : 1];
            o[0] += 1.0; o[1] += ms; o[2] += r.flops; o[3] += r.bytes;
            c->prof_retired++;
        } else {
            (void)hipGetLastError();
            c->prof_lost++;
        }
        c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b);
        c->prof_recs.pop_front();
    }
}

int ctx_gemm(cr_ctx* c, int epi, const GemmParams& p, hipStream_t st) {
    cr_ctx::ProfRec rec{};
    bool on = c->prof && (c->prof_mode != 2 || p.M >= 1024);
    if (on) {
        if (c->prof_recs.size() >= PROF_SOFT) prof_retire(c, false);
        hipEvent_t* ev[2] = {&rec.a, &rec.b};
        for (int i = 0; i < 2 && on; i++) {
            if (!c->prof_pool.empty()) { *ev[i] = c->prof_pool.back(); c->prof_pool.pop_back(); }
            else if (hipEventCreate(ev[i]) != hipSuccess) {
                (void)hipGetLastError();
                if (i == 1) c->prof_pool.push_back(rec.a);       // keep the first event for the next launch
                c->prof_lost++;
                on = false;                                     // measure nothing for this launch, but do launch it
            }
        }
        if (on) hipEventRecord(rec.a, st);
    }
    const int r = launch_gemm(epi, p, st);
    if (on) {
        hipEventRecord(rec.b, st);
        const double n_out = epi == EPI_SWIGLU ? p.N / 2.0 : p.N;
        rec.flops = 2.0 * p.M * (double)p.N * p.K;
        rec.bytes = (p.w8 ? 1.0 : 2.0) * (double)p.N * p.K + (p.a8 ? 1.0 : 2.0) * (double)p.M * p.K + (p.a8 ? 4.0 * (p.M + p.N) : 0.0) + (epi == EPI_ARGMAX ? 8.0 * p.M * ((p.N + 63) / 64) : (epi == EPI_F32 ? 4.0 : 2.0) * p.M * n_out);
        rec.big = p.M >= 1024;
        c->prof_recs.push_back(rec);
        c->prof_issued++;
        if ((int64_t)c->prof_recs.size() > c->prof_peak_pending) c->prof_peak_pending = (int64_t)c->prof_recs.size();
    }
    if (r != CR_OK) return cr_fail(r, "gemm(epi=%d M=%d N=%d K=%d) rejected or failed to launch", epi, p.M, p.N, p.K);
    return CR_OK;
}

static size_t dtype_size(int dt) { return dt == CR_BF16 ? 2 : dt == CR_F32 ? 4 : dt == CR_I64 ? 8 : dt == CR_I32 ? 4 : 0; }   // CR_U8 is internal: not loadable

int ctx_share_ok(const cr_ctx* c, const char* who) {
    if (c->borrowed && c->owner_cell && c->owner_cell->load(std::memory_order_acquire) != c->owner_gen)
        return cr_fail(CR_ERR_STATE, "%s: the context that owns these weights was destroyed, reloaded, re-finalized or switched an fp8 option since cr_share_weights: share again", who);
    return CR_OK;
}

extern "C" {

const char* cr_last_error(void) { return g_err; }
int cr_abi_version(void) { return CR_ABI_VERSION; }
#ifndef CR_BUILD_ID
#define CR_BUILD_ID "CR_BUILD_ID=unknown"
#endif
// the literal keeps its "CR_BUILD_ID=" tag so that build.py can find the id in the file without loading it
const char* cr_build_id(void) { static const char id[] = CR_BUILD_ID; return id + 12; }

// csrc/diag.hpp: translation units compiled with a diagnostic macro announce it here at load time (static constructors); "" for the product build
static std::string& diag_flags() { static std::string s; return s; }
int cr_diag_register(const char* flags) { if (flags && *flags) diag_flags() += flags; return 0; }
const char* cr_build_flags(void) { return diag_flags().c_str(); }

int cr_create(int device, const cr_model_desc* desc, cr_ctx** out) {
    if (!desc || !out) return cr_fail(CR_ERR_ARG, "cr_create: null argument");
    CR_HIP(hipSetDevice(device));
    cr_ctx* c = new cr_ctx();
    c->device = device;
    c->d = *desc;
    { const char* e = getenv("CR_NO_SLICED_DECODE"); c->no_sliced_decode = e && atoi(e) != 0; }
    { const char* e = getenv("CR_DECODE_GRAPH"); if (e) c->decode_graph = atoi(e) != 0; }
    { const char* e = getenv("CR_DECODE_FUSED"); if (e) c->fused_decode = atoi(e) != 0; }
    { const char* e = getenv("CR_DECODE_FOLD_ROPE"); if (e) c->fold_rope = atoi(e) != 0; }
    { const char* e = getenv("CR_PERCEIVER_ATTN_V1"); if (e) c->perceiver_v1 = atoi(e) != 0; }
    { const char* e = getenv("CR_PREFILL_LAST_ROWS"); if (e) c->prefill_last_rows = atoi(e) != 0; }
    c->scratch_bytes = 1 << 20;
    if (hipMalloc((void**)&c->scratch, c->scratch_bytes) != hipSuccess) { delete c; return cr_fail(CR_ERR_NOMEM, "scratch"); }
    hipMemset(c->scratch, 0, c->scratch_bytes);
    *out = c;
    return CR_OK;
}

int cr_destroy(cr_ctx* c) {
    if (!c) return CR_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    c->gen_cell->store(~(uint64_t)0, std::memory_order_release);      // borrowers of this context's tensors now fail ctx_share_ok instead of reading freed memory
    if (!c->borrowed) for (auto& kv : c->w) if (kv.second.ptr) hipFree(kv.second.ptr);
    if (c->ws) hipFree(c->ws);
    if (c->scratch) hipFree(c->scratch);
    if (c->side) hipStreamDestroy(c->side);
    for (auto& r : c->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto e : c->prof_pool) hipEventDestroy(e);
    delete c;
    return CR_OK;
}

int cr_share_weights(cr_ctx* dst, const cr_ctx* src) {
    if (!dst || !src || dst == src) return cr_fail(CR_ERR_ARG, "cr_share_weights: bad argument");
    if (dst->device != src->device) return cr_fail(CR_ERR_ARG, "cr_share_weights: contexts on different devices");
    if (!dst->w.empty() && !dst->borrowed) return cr_fail(CR_ERR_STATE, "cr_share_weights: the destination owns weights of its own");
    if (!src->finalized) return cr_fail(CR_ERR_STATE, "cr_share_weights: finalize the source first");
    if (ctx_share_ok(src, "cr_share_weights (source)") != CR_OK) return CR_ERR_STATE;      // a stale borrower hands on nothing
    dst->w = src->w;                    // device pointers only: nothing is copied
    dst->borrowed = true;
    dst->d = src->d;
    dst->finalized = true;
    dst->fp8_decode = src->fp8_decode; dst->fp8_mfma = src->fp8_mfma;
    // the cell of the context that owns the tensors: the source's own, or (source = a borrower) the one the source watches
    dst->owner_cell = src->borrowed ? src->owner_cell : src->gen_cell;
    dst->owner_gen = src->borrowed ? src->owner_gen : src->weight_gen;
    cr_bump_gen(dst);
    return CR_OK;
}

int cr_load_weight(cr_ctx* c, const char* name, const void* src, int dtype, const int64_t* shape, int ndim,
                   int src_is_host, void* stream) {
    if (!c || !name || !src || !shape || ndim <= 0 || ndim > 8) return cr_fail(CR_ERR_ARG, "cr_load_weight: bad argument");
    if (c->borrowed) return cr_fail(CR_ERR_STATE, "cr_load_weight(%s): this context borrows its weights (cr_share_weights); load into the owner", name);
    const size_t es = dtype_size(dtype);
    if (!es) return cr_fail(CR_ERR_ARG, "cr_load_weight(%s): unknown dtype %d", name, dtype);
    DevTensor t;
    t.dtype = dtype;
    t.shape.assign(shape, shape + ndim);
    if (t.numel() <= 0) return cr_fail(CR_ERR_ARG, "cr_load_weight(%s): empty tensor", name);
    t.bytes = (size_t)t.numel() * es;
    auto it = c->w.find(name);
    if (it != c->w.end()) {
        if (it->second.bytes != t.bytes) { hipFree(it->second.ptr); c->w.erase(it); it = c->w.end(); }
        else t.ptr = it->second.ptr;
    }
    if (!t.ptr && hipMalloc(&t.ptr, (t.bytes + 255) & ~(size_t)255) != hipSuccess)
        return cr_fail(CR_ERR_NOMEM, "cr_load_weight(%s): %zu bytes", name, t.bytes);
    CR_HIP(hipMemcpyAsync(t.ptr, src, t.bytes, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (src_is_host) CR_HIP(hipStreamSynchronize((hipStream_t)stream));   // caller may free host memory right away
    c->w[name] = t;
    // copies cr_finalize / cr_enable_fp8_* derived from the tensor this one replaces (decode layout, e4m3 bytes + scales, fc1's bound) would go on
    // serving the OLD values to the kernels that read them (round-4 advice): they go now, cr_finalize rebuilds the decode layout, and the fp8
    // options switch off until cr_enable_fp8_* is called again (it rebuilds what is missing)
    bool had_fp8 = false;
    for (const char* pre : {"declayout.", "fp8.", "fp8s.", "fp8b.", "fp8dl."}) {
        auto d = c->w.find(std::string(pre) + name);
        if (d == c->w.end()) continue;
        if (pre[0] == 'f') had_fp8 = true;
        hipFree(d->second.ptr);
        c->w.erase(d);
    }
    {   // fc1's bound {max row norm of W1, max |b1|} also depends on the bias
        const std::string nm(name);
        const size_t at = nm.rfind("mlp.fc1.bias");
        if (at != std::string::npos && at + 12 == nm.size()) {
            auto d = c->w.find("fp8b." + nm.substr(0, at) + "mlp.fc1.weight");
            if (d != c->w.end()) { hipFree(d->second.ptr); c->w.erase(d); had_fp8 = true; }
        }
    }
    if (had_fp8) { c->fp8_decode = false; c->fp8_mfma = 0; }
    cr_bump_gen(c);
    if (strncmp(name, "orderformer.", 12) != 0) c->finalized = false;      // the sorter (f4) has no derived tensors to refresh
    return CR_OK;
}

int cr_profile(cr_ctx* c, int enable) {
    if (!c) return cr_fail(CR_ERR_ARG, "cr_profile: null context");
    c->prof = enable != 0;
    c->prof_mode = enable;
    return CR_OK;
}

int cr_profile_read(cr_ctx* c, double* out) {
    if (!c || !out) return cr_fail(CR_ERR_ARG, "cr_profile_read: null argument");
    CR_HIP(hipSetDevice(c->device));
    prof_retire(c, true);
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < 4; i++) { out[k * 4 + i] = c->prof_acc[k][i]; c->prof_acc[k][i] = 0.0; }
    return CR_OK;
}

int cr_profile_stats(cr_ctx* c, int64_t* out) {
    if (!c || !out) return cr_fail(CR_ERR_ARG, "cr_profile_stats: null argument");
    out[0] = c->prof_issued; out[1] = c->prof_retired + (int64_t)c->prof_recs.size(); out[2] = c->prof_lost; out[3] = c->prof_peak_pending;
    return CR_OK;
}

// ---- single operators ---------------------------------------------------------------------------
int cr_op_gemm(int epi, const void* A, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc,
               const void* bias, const void* scale, const void* res, int64_t ldr, int M, int N, int K, int group,
               void* stream) {
    GemmParams p{};
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)Wt; p.ldw = ldw; p.C = C; p.ldc = ldc;
    p.bias = (const bf16*)bias; p.scale = (const bf16*)scale; p.res = (const bf16*)res; p.ldr = ldr;
    p.M = M; p.N = N; p.K = K; p.group = group;
    const int kern = (epi >> 8) & 0xff;           // tests pin a kernel: 1 = 128x128, 2 = 256x256, 3 = skinny
    p.kernel = kern == 1 ? 128 : (kern == 2 || kern == 5 || kern == 6) ? 256 : kern == 3 ? 1 : 0;
    p.slots = kern == 5 ? 16 : kern == 6 ? 32 : 0;            // 5 / 6: the 256x256 kernel with its 16- / 32-MFMA-slot schedule pinned
    if (epi & (1 << 16)) { p.w8 = 1; p.wscale = (const float*)scale; p.scale = nullptr; }      // e4m3 weights + per-row fp32 scales
    if (epi & (1 << 17)) { p.a8 = 1; p.ascale = (const float*)res; p.res = nullptr; }          // e4m3 activations too: `res` = fp32 row scales [M]
    p.wsw = (epi >> 18) & 3;                      // weights in the decode layout (cr_op_decode_swizzle): 1 plain tiles, 2 wqkv's RoPE tile order
    epi &= 0xff;
    int r = launch_gemm(epi, p, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_gemm(epi=%d, M=%d, N=%d, K=%d) rejected or failed to launch", epi, M, N, K);
    return CR_OK;
}

int cr_op_gemm_q8(const void* a8, const float* ascale, const void* w8, const float* wscale, const void* bias, const float* c8scale,
                  void* c8, int M, int N, int K, void* stream) {
    GemmParams p{};
    p.A = (const bf16*)a8; p.lda = K; p.W = (const bf16*)w8; p.ldw = K; p.C = c8; p.ldc = N; p.bias = (const bf16*)bias; p.M = M; p.N = N; p.K = K;
    p.w8 = 1; p.wscale = wscale; p.a8 = 1; p.ascale = ascale; p.c8scale = c8scale;
    const int r = launch_gemm(EPI_GELU_Q8, p, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_gemm_q8(M=%d, N=%d, K=%d) rejected or failed to launch", M, N, K);
    return CR_OK;
}

int cr_op_layernorm(const void* in, void* out, const void* gamma, const void* beta, int64_t rows, int n, float eps,
                    int pixel_shuffle, void* stream) {
    NormParams p{};
    p.in = (const bf16*)in; p.ld_in = n; p.out = (bf16*)out; p.ld_out = n;
    p.gamma = (const bf16*)gamma; p.beta = (const bf16*)beta; p.rows = rows; p.eps = eps;
    int r = launch_layernorm(p, n, pixel_shuffle ? 1 : 0, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_layernorm(rows=%lld, n=%d) rejected or failed", (long long)rows, n);
    return CR_OK;
}

int cr_op_norm_fp8(const void* in, const void* gamma, const void* beta, int64_t rows, int n, float eps, void* out8, float* out_scale,
                   float* next_scale, const float* next_bound, void* stream) {
    if (!in || !gamma || !out8 || !out_scale || ((next_scale != nullptr) != (next_bound != nullptr))) return cr_fail(CR_ERR_ARG, "cr_op_norm_fp8: bad argument");
    NormParams p{};
    p.next_scale = next_scale; p.next_bound = next_bound;
    p.in = (const bf16*)in; p.ld_in = n; p.out = nullptr; p.ld_out = n; p.gamma = (const bf16*)gamma; p.beta = (const bf16*)beta;
    p.rows = rows; p.eps = eps; p.out8 = (unsigned char*)out8; p.out8_scale = out_scale;
    const int r = beta ? launch_layernorm(p, n, 0, (hipStream_t)stream) : launch_rmsnorm(p, n, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_norm_fp8(rows=%lld, n=%d) rejected or failed", (long long)rows, n);
    return CR_OK;
}

int cr_op_rmsnorm(const void* in, void* out, const void* gamma, int64_t rows, int n, float eps, void* stream) {
    NormParams p{};
    p.in = (const bf16*)in; p.ld_in = n; p.out = (bf16*)out; p.ld_out = n;
    p.gamma = (const bf16*)gamma; p.rows = rows; p.eps = eps;
    int r = launch_rmsnorm(p, n, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_rmsnorm(rows=%lld, n=%d) rejected or failed", (long long)rows, n);
    return CR_OK;
}

int cr_op_decode_gemm(int which, int flags, const void* W, int64_t ldw, int M, int N, int K, const void* X, int64_t ldx, const void* xres,
                      const void* gamma, float eps, void* xio, void* C, int64_t ldc, const void* cosT, const void* sinT, void* q_out, void* kc,
                      void* vc, const int32_t* seqs, const int32_t* lens, int max_tokens, void* stream) {
    DecodeGemmParams p{};
    p.W = (const bf16*)W; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.X = (const bf16*)X; p.ldx = ldx; p.xres = (const bf16*)xres;
    p.gamma = (const bf16*)gamma; p.eps = eps; p.xio = (bf16*)xio; p.C = C; p.ldc = ldc; p.cosT = (const bf16*)cosT; p.sinT = (const bf16*)sinT;
    p.q_out = (bf16*)q_out; p.kc = (bf16*)kc; p.vc = (bf16*)vc; p.seqs = seqs; p.lens = lens; p.max_tokens = max_tokens; p.flags = flags & 255;
    p.swizzled = (flags >> 8) & 1;
    const int r = launch_decode_gemm(which, p, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_decode_gemm(which=%d, M=%d, N=%d, K=%d) rejected or failed to launch", which, M, N, K);
    return CR_OK;
}

int cr_op_decode_swizzle(int which, const void* W, int64_t ldw, int N, int K, void* dst, void* stream) {
    // which = 8: e4m3 bytes [N][ldw] (cr_enable_fp8_decode's copies; K % 64 == 0), else bf16 for op_decode_gemm's `which`
    const int r = which == 8 ? decode_swizzle_weight8((const unsigned char*)W, ldw, N, K, (unsigned char*)dst, (hipStream_t)stream)
                             : decode_swizzle_weight(which, (const bf16*)W, ldw, N, K, (bf16*)dst, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_decode_swizzle(which=%d, N=%d, K=%d) rejected or failed to launch", which, N, K);
    return CR_OK;
}

int cr_op_attention(const void* q, const void* k, const void* v, void* o, const int64_t* s, int B, int H,
                    int Sq, int Sk, int head_dim, int kv_group, int causal, int q_pos0, float q_prescale, float s_div,
                    void* stream) {
    if (!s) return cr_fail(CR_ERR_ARG, "cr_op_attention: strides");
    AttnParams p{};
    p.Q = (const bf16*)q; p.K = (const bf16*)k; p.V = (const bf16*)v; p.O = (bf16*)o;
    p.q_bs = s[0]; p.q_rs = s[1]; p.q_hs = s[2];
    p.k_bs = s[3]; p.k_rs = s[4]; p.k_hs = s[5];
    p.v_bs = s[6]; p.v_rs = s[7]; p.v_hs = s[8];
    p.o_bs = s[9]; p.o_rs = s[10]; p.o_hs = s[11];
    p.B = B; p.H = H; p.Sq = Sq; p.Sk = Sk; p.kv_group = kv_group; p.q_pos0 = q_pos0;
    p.q_prescale = q_prescale; p.s_div = s_div;
    if (!causal && head_dim == 64 && Sq == Sk && Sq > 256 && ((Sq - 1) & 127) == 0 && s_div == 1.0f && kv_group == 1) {
        // the ViT layout takes attention_vit.hip, which wants scratch for the CLS query's partials: this test / profiling entry point
        // keeps one grow-only buffer per DEVICE (stage entry points carve theirs out of the context's workspace); growing waits for the
        // device first, and the lock makes concurrent callers take turns (the kernels of one device's callers share the buffer: this
        // entry point is for one caller per device at a time, as the header says)
        static std::mutex mu;
        static float* scratch[64] = {};
        static size_t scratch_n[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return cr_fail(CR_ERR_HIP, "cr_op_attention: device");
        const size_t need = vit_attn_ws_floats(B, H, Sq);
        std::lock_guard<std::mutex> lk(mu);
        if (need > scratch_n[dev]) {
            hipDeviceSynchronize();
            if (scratch[dev]) hipFree(scratch[dev]);
            scratch[dev] = nullptr; scratch_n[dev] = 0;
            if (hipMalloc((void**)&scratch[dev], need * 4) != hipSuccess) return cr_fail(CR_ERR_NOMEM, "cr_op_attention: scratch");
            scratch_n[dev] = need;
        }
        p.part_ml = scratch[dev];
    }
    int r = launch_flash_attn(p, head_dim, causal != 0, (hipStream_t)stream);
    if (r != CR_OK) return cr_fail(r, "cr_op_attention(d=%d) rejected or failed", head_dim);
    return CR_OK;
}

int64_t cr_op_decode_attention_scratch_floats(int B, int max_keys) {
    const int nsplit = ((max_keys + 63) / 64 + ATTN_SPLIT_TILES - 1) / ATTN_SPLIT_TILES;
    return (int64_t)attn_split_ws_floats(B, 8, 4, nsplit, 128);
}

int cr_op_decode_attention(int which, const void* q, const void* kc, const void* vc, int max_tokens, const int32_t* seqs, const int32_t* lens, int B, int max_keys,
                           float s_div, float* scratch, void* out, void* stream) {
    if (!q || !kc || !vc || !seqs || !lens || !scratch || !out || B <= 0 || max_keys <= 0 || max_tokens < max_keys || (which != 0 && which != 1))
        return cr_fail(CR_ERR_ARG, "cr_op_decode_attention: bad argument");
    constexpr int HD = 128, NKV = 8, G = 4, D = NKV * G * HD;
    AttnParams ap{};
    ap.K = (const bf16*)kc; ap.V = (const bf16*)vc; ap.Q = (const bf16*)q; ap.O = (bf16*)out;
    ap.k_bs = ap.v_bs = (int64_t)NKV * max_tokens * HD; ap.k_rs = ap.v_rs = HD; ap.k_hs = ap.v_hs = (int64_t)max_tokens * HD;
    ap.q_prescale = 1.0f; ap.s_div = s_div;
    ap.q_bs = D; ap.q_rs = HD; ap.q_hs = G * HD; ap.o_bs = D; ap.o_rs = HD; ap.o_hs = G * HD;
    ap.B = B; ap.H = NKV; ap.Sq = G; ap.Sk = 0; ap.kv_group = 1; ap.q_pos0 = 0;
    ap.seq_map = seqs; ap.sk_arr = lens; ap.sk_add = 1;
    ap.nsplit = ((max_keys + 63) / 64 + ATTN_SPLIT_TILES - 1) / ATTN_SPLIT_TILES;
    ap.part_ml = scratch; ap.part_o = scratch + (size_t)B * NKV * ap.nsplit * G * 2;
    ap.force_matrix_core = which == 1;                    // launch_flash_attn_split keeps the matrix-core split kernel, as CR_DECODE_ATTN=0 does for a whole process
    if (launch_flash_attn_split(ap, HD, (hipStream_t)stream) != CR_OK) return cr_fail(CR_ERR_HIP, "cr_op_decode_attention: launch failed");
    return CR_OK;
}

}  // extern "C"
