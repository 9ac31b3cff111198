/* callireader_hip.h — C ABI of libcallireader_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for CalliReader's image->text hot path.  The reference is pure
 * Python; each entry point replaces the PyTorch module call named next to it
 * (paths relative to the reference repo).  The Python binding a maintainer would
 * add is shown in INTEGRATION.md and shipped as callireader_amd/_binding.py.
 *
 * Conventions
 *   - every pointer argument is a DEVICE pointer owned by the caller unless it is
 *     documented as host; buffers are dense row-major bf16 unless stated;
 *   - every launch takes a hipStream_t (passed as void*); calls only enqueue work,
 *     except cr_create/cr_destroy/cr_load_weight/cr_finalize/cr_kv_alloc/cr_kv_free,
 *     which may allocate and synchronise;
 *   - return value: 0 = CR_OK, negative = error; cr_last_error() returns a
 *     thread-local, NUL-terminated description of the last failure;
 *   - no exceptions cross the ABI, no internal threads, one context per device,
 *     a context is not re-entrant.
 */
#ifndef CALLIREADER_HIP_H
#define CALLIREADER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CR_ABI_VERSION 10  /* 10: cr_op_decode_attention(+_scratch_floats); 9: cr_build_flags, cr_diag_register; 8: cr_op_decode_swizzle, cr_op_decode_gemm flags bit 8, cr_op_gemm bits 18-19; 7: cr_op_decode_gemm; 6: cr_build_id, cr_llm_hidden_probe; 5: cr_share_weights, cr_op_gemm_q8, cr_op_norm_fp8 takes the next linear's bound; 4: cr_enable_fp8_mfma, cr_op_norm_fp8, cr_op_gemm bit 17; 2: cr_orderformer; cr_op_gemm kernel pin and EPI_PARTIAL (epi 7); 3: cr_profile_stats, cr_kv_read, cr_kv_reset takes a stream, cr_enable_fp8_decode, cr_op_quantize_fp8, epi 8 */

enum { CR_OK = 0, CR_ERR_ARG = -1, CR_ERR_HIP = -2, CR_ERR_STATE = -3, CR_ERR_NOMEM = -4 };
enum { CR_BF16 = 0, CR_F32 = 1, CR_I64 = 2, CR_I32 = 3, CR_U8 = 4 /* library-internal: e4m3 weight copies */ };

typedef struct cr_ctx cr_ctx;
typedef struct cr_kv cr_kv;

/* Shapes: InternVL/config.json (vision_config :114-143, llm_config :14-103, downsample_ratio :11),
 * models/perceiver_resampler.py:54-64, InternVL/modeling_internvl_chat.py:157. */
typedef struct cr_model_desc {
    int32_t vit_layers;     /* 24  */
    int32_t rs_depth;       /* 4   */
    int32_t llm_layers;     /* 32  */
    int32_t vocab;          /* 92553 */
    int32_t max_pos;        /* rows of the RoPE tables handed over by the host (<= 32768) */
    float vit_ln_eps;       /* 1e-6 */
    float rms_eps;          /* 1e-5 */
    int32_t reserved[8];
} cr_model_desc;

const char* cr_last_error(void);
int cr_abi_version(void);
/* sha256 (first 16 hex digits) of the .hip / .hpp sources and of this header the library was compiled from, set by
 * callireader_amd/build.py (-DCR_BUILD_ID): lets a log prove which source tree a run used.  "unknown" for a hand build. */
const char* cr_build_id(void);
/* Names of the diagnostic macros any translation unit of this library was compiled with (callireader_amd/csrc/diag.hpp: knock-out / poison / stamp builds
 * made by scripts/build_variant.py, several of which give wrong results by design), space-separated; "" for the product build.  The ctypes binding refuses a
 * library that reports any unless CR_HIP_LIB names it explicitly.  cr_diag_register is how such a translation unit announces itself (load time). */
const char* cr_build_flags(void);
int cr_diag_register(const char* flags);

/* ---- lifetime & weights --------------------------------------------------------------------- */
/* InternVLChatModel.__init__ (InternVL/modeling_internvl_chat.py:136-194) */
int cr_create(int device, const cr_model_desc* desc, cr_ctx** out);
int cr_destroy(cr_ctx* ctx);
/* Two stages in flight (the batched decode of one batch of pages beside the visual stage and prefill of the next; the reference runs
 * page after page, modeling_internvl_chat.py:649-762 -- a throughput arrangement above the same per-page results).  A context is
 * not re-entrant: its stage entry points carve their activations out of one grow-only workspace.  cr_share_weights makes `dst`
 * (fresh from cr_create with the same description) use the weights, derived tensors and fp8 copies of the finalized `src` WITHOUT
 * copying them: a second host thread can then run stages through `dst` on its own stream (own workspace, own profiler) while
 * the first works through `src`.  KV caches (cr_kv_alloc) are plain device state and may be filled through one context and read
 * through the other; ordering between the two streams is the caller's (events).  `src` must outlive `dst`; call again after
 * reloading weights or toggling the fp8 options on `src`. */
int cr_share_weights(cr_ctx* dst, const cr_ctx* src);
/* One call per checkpoint tensor; `name` is the safetensors key (InternVL/model.safetensors.index.json), or
 *   "calli.mu" / "calli.sigma"  (vocab,1)  — params/gauss_norm_mu_sigma.pth columns (modeling_internvl_chat.py:153-155)
 *   "rope.cos" / "rope.sin"     (max_pos,128) bf16 — InternLM2DynamicNTKScalingRotaryEmbedding cache
 *                                (InternVL/modeling_internlm2.py:213-229), built on the host exactly as the reference does.
 * The library copies (and may re-layout) the data; `src_is_host` selects H2D vs D2D. */
int cr_load_weight(cr_ctx* ctx, const char* name, const void* src, int dtype, const int64_t* shape, int ndim,
                   int src_is_host, void* stream);
/* Build derived tensors once all weights are in (K-padded patch kernel, interleaved w1|w3, L2-normalised VQ table). */
int cr_finalize(cr_ctx* ctx, void* stream);

/* ---- vision ---------------------------------------------------------------------------------- */
/* InternVisionModel.forward(pixel_values).last_hidden_state — InternVL/modeling_intern_vit.py:399-437
 * pixels [T,3,448,448] -> out [T,1025,1024] */
int cr_vit_forward(cr_ctx* ctx, const void* pixels, int T, void* out, void* stream);
/* drop CLS + pixel_shuffle(0.5, v2) + mlp1 — InternVL/modeling_internvl_chat.py:283-297,311-318
 * vit_out [T,1025,1024] -> out [T,256,4096] */
int cr_project(cr_ctx* ctx, const void* vit_out, int T, void* out, void* stream);
/* InternVLChatModel.extract_feature — InternVL/modeling_internvl_chat.py:299-319 (= the two calls above) */
int cr_extract_feature(cr_ctx* ctx, const void* pixels, int T, void* out, void* stream);

/* ---- tile preprocessing (host side of the reference: utils/utils.py:354-478) ------------------------------- */
/* One resize job: crop [sx0,sy0,sw,sh] of the page, Pillow Image.resize((ow,oh)) (default BICUBIC, antialiased),
 * then either mode 0: paste at (left,top) on a white 448x448 canvas = ONE tile `tile0` (load_image_2, :420-452), or
 * mode 1: cut into 448x448 tiles tile0.. row-major, `cols` per row (dynamic_preprocess, :381-417; cols = ow/448). */
typedef struct cr_prep_job {
    int32_t sx0, sy0, sw, sh, ow, oh, mode, tile0, cols, left, top;
} cr_prep_job;
/* page_rgb: device uint8 [H][W][3]; jobs: host array; lut: device bf16 [3][256] = bf16((p/255 - mean_c)/std_c) built by
 * the host with the reference's fp32 expression (build_transform, :354-362); out_tiles: device bf16 [n_tiles,3,448,448].
 * Bit-identical to the reference's PIL/torch pipeline (tests/test_gpu_prep.py). */
int cr_preprocess(cr_ctx* ctx, const void* page_rgb, int H, int W, const cr_prep_job* jobs, int n_jobs, const void* lut,
                  void* out_tiles, int n_tiles, void* stream);

/* ---- CalliAlign ------------------------------------------------------------------------------- */
/* PerceiverResampler.forward — models/perceiver_resampler.py:81-100:  in [T,256,4096] -> out [T,3,4096] */
int cr_resample(cr_ctx* ctx, const void* in, int T, void* out, void* stream);
/* vq_cos_sim — models/similarity.py:9-27:  in [n,4096] -> idx [n] int64, cos [n] bf16 (cos may be NULL) */
int cr_vq(cr_ctx* ctx, const void* in, int n, int64_t* idx, void* cos, void* stream);
/* calli_align tail — InternVL/modeling_internvl_chat.py:602-640.
 * flags bit0 = drop_zero, bit1 = hard_vq (needs cos).  out [n,4096]; *n_out (device int32) = rows kept. */
int cr_denorm(cr_ctx* ctx, const void* in, const int64_t* idx, const void* cos, int n, int flags,
              void* out, int32_t* n_out, void* stream);

/* ---- language model --------------------------------------------------------------------------- */
/* generate_ocr / generate_origin head — InternVL/modeling_internvl_chat.py:1081-1107,1033-1052:
 * out[s] = tok_embeddings[ids[s]], rows with ids == img_id overwritten by vit_embeds (in order),
 * rows with ids == ref_id by ref_embeds.  n_vit / n_ref rows must equal the id counts (checked on the host side). */
int cr_embed_splice(cr_ctx* ctx, const int64_t* ids, int S, const void* vit_embeds, int n_vit, int64_t img_id,
                    const void* ref_embeds, int n_ref, int64_t ref_id, void* out, void* stream);

/* KV cache for `n_seqs` sequences of up to `max_tokens` tokens: replaces the reference's tuple cache grown by
 * torch.cat (InternVL/modeling_internlm2.py:383-388).  The object also keeps, per sequence, the ids generated so
 * far — what transformers' generate() carries as `input_ids` when only inputs_embeds is given — so that the
 * repetition penalty and the next step's input never leave the device. */
int cr_kv_alloc(cr_ctx* ctx, int n_seqs, int max_tokens, cr_kv** out);
int cr_kv_free(cr_kv* kv);
int cr_kv_length(const cr_kv* kv, int seq);           /* tokens currently cached (host-side bookkeeping) */
/* Forget sequence `seq` (length 0, no generated ids), or every sequence when seq < 0.  Enqueued on `stream`, no
 * device-wide synchronisation: ordered after earlier work on that stream and before what the caller enqueues next. */
int cr_kv_reset(cr_kv* kv, int seq, void* stream);
/* One cached position as the reference's tuple cache holds it (InternVL/modeling_internlm2.py:383-388, K after RoPE):
 * out [8 kv heads][128] bf16 (device) = past_key_values[layer][which][seq, :, pos, :]; which 0 = K, 1 = V.  Parity tests. */
int cr_kv_read(cr_kv* kv, int layer, int seq, int pos, int which, void* out, void* stream);
/* Copy the ids generated so far for `seq` into host memory (at most `max`); returns the count, or < 0.
 * Synchronises `stream`. */
int cr_kv_generated(cr_kv* kv, int seq, int64_t* out_host, int max, void* stream);

/* InternLM2ForCausalLM.forward(inputs_embeds=…, use_cache=True) — InternVL/modeling_internlm2.py:1022-1110 —
 * for ONE sequence `seq`, appended at its current length; then the first greedy pick of
 * transformers 4.45.2 GenerationMixin._sample (RepetitionPenaltyLogitsProcessor over the ids generated so far,
 * argmax with first-max-wins), appended to the sequence's generated ids.
 * embeds [S,4096]; last_logits [vocab] fp32 = raw last-row logits before the penalty (may be NULL). */
int cr_llm_prefill(cr_ctx* ctx, cr_kv* kv, int seq, const void* embeds, int S, float penalty, float* last_logits,
                   void* stream);
/* The same for n sequences at once: embeds is the concatenation [sum(lens), 4096] of the prompts in seqs[] order
 * (seqs, lens: host arrays).  The linear layers run over all prompt rows together (better MFMA tile occupancy),
 * attention stays per sequence; every sequence gets exactly the result of its own cr_llm_prefill.
 * last_logits [n,vocab] fp32 (may be NULL). */
int cr_llm_prefill_batch(cr_ctx* ctx, cr_kv* kv, const int32_t* seqs, int n, const void* embeds, const int32_t* lens,
                         float penalty, float* last_logits, void* stream);
/* Parity tooling: while `dst` is not NULL, every cr_llm_prefill / cr_llm_prefill_batch also copies rows [row0, row0 + rows) of the
 * residual stream (row index into the concatenated prompt rows) into dst [llm_layers + 1][rows][4096] bf16 (device): slot 0 = the
 * decoder stack's input, slot l + 1 = the output of InternLM2DecoderLayer l (InternVL/modeling_internlm2.py:621-681; what
 * output_hidden_states=True collects at :916-918,965-967, before the final norm).  dst = NULL switches it off. */
int cr_llm_hidden_probe(cr_ctx* ctx, void* dst, int row0, int rows);
/* One greedy step for sequences seqs[0..n) (host array): embed each sequence's last generated id (or
 * force_tokens[i], device int64, when not NULL), run the decoder against the cache, append K/V, apply the
 * penalty, argmax, append the new id.  logits [n,vocab] fp32 raw (may be NULL).  Replaces one iteration of
 * GenerationMixin._sample as driven from InternVL/modeling_internvl_chat.py:1111-1120 with
 * prepare_inputs_for_generation (InternVL/modeling_internlm2.py:1112-1149). */
int cr_llm_decode(cr_ctx* ctx, cr_kv* kv, const int32_t* seqs, int n, const int64_t* force_tokens, float penalty,
                  float* logits, void* stream);

/* fp8 weight path (BASELINE.json configs[4]; the reference itself has no fp8 -- an option, OFF by default, the headline stays
 * bf16).  enable != 0: every linear weight of the language model the batched decode streams (wqkv, wo, w1|w3, w2 of each
 * layer, the LM head) gets an e4m3 (OCP e4m3fn) copy with one fp32 scale per output row, scale = max|w| / 448, built on the
 * device on first use; cr_llm_decode with <= 64 sequences then reads those (half the HBM bytes), dequantising in registers
 * (exact: e4m3 is a subset of bf16), multiplying in the same bf16 MFMA with fp32 accumulation and applying the row scale to
 * the fp32 sum.  Activations, KV cache, prefill and everything visual stay bf16.  enable == 0 switches back (copies kept).
 * Call after cr_finalize; re-run it after reloading weights. */
int cr_enable_fp8_decode(cr_ctx* ctx, int enable, void* stream);
/* fp8 on the matrix cores (same standing: an option, OFF by default; a THROUGHPUT option -- e4m3 keeps 3 mantissa bits, every such
 * linear adds ~5 % of relative noise to its output, and whether the transcription survives that is what evaluate.py --compare_fp8 measures
 * on a real checkpoint; nothing here is claimed to preserve the reference's tokens).
 * enable = 1: the linears whose input is a norm's output -- QKV and fc1 of every ViT layer, mlp1's first linear, wqkv and w1|w3 of every
 *   LLM layer in PREFILL -- get the e4m3 copy + row scale above, the norm kernel in front of each writes the normalised row as e4m3 with
 *   one fp32 scale per row (max|y| / 448 over the bf16-rounded row) instead of bf16, and the 256x256 tiled kernel multiplies e4m3 x e4m3
 *   with v_mfma_f32_16x16x128_f8f6f4 (fp32 accumulation, twice the bf16 rate), applying ascale[m] * wscale[n] to the finished sum before
 *   the bias and the rest of the epilogue.  proj / fc2 / wo / w2, attention, the residual stream, the KV cache and decode stay bf16.
 * enable = 2: additionally the linears whose input no norm produces: ViT fc2 (fc1's epilogue writes its GELU output as e4m3 rows under a
 *   LayerNorm-derived bound) and the LLM's wo / w2 in prefill (one quantiser pass over their input each).
 * enable = 0 switches back (copies kept).  Call after cr_finalize; re-run it after reloading weights. */
int cr_enable_fp8_mfma(cr_ctx* ctx, int enable, void* stream);

/* ---- measurement ------------------------------------------------------------------------------- */
/* While enabled, every launch of the dense-GEMM kernel made by the stage entry points is bracketed by a pair of
 * HIP events on the launch stream.  cr_profile_read synchronises and returns, for the compute-bound class
 * (M >= 1024 rows) in out[0..3] = {launches, summed kernel ms, summed algorithmic FLOPs (2*M*N*K), algorithmic bytes} and for the
 * weight-streaming class (M < 1024) in out[4..7] = {launches, ms, FLOPs, algorithmic bytes (W + A + C)}; then clears. */
/* enable: 0 off, 1 every GEMM launch, 2 only the compute-bound class (batched decode then runs as a captured hipGraph,
 * whose launches cannot carry events). */
int cr_profile(cr_ctx* ctx, int enable);
int cr_profile_read(cr_ctx* ctx, double* out8);
/* Bookkeeping of the above since cr_create: out[0] = launches bracketed, out[1] = launches accounted for (retired into
 * the sums or still pending), out[2] = launches whose events could not be created or read (0 in a healthy run),
 * out[3] = most records ever pending at once.  Records are retired as their events complete; there is no cap. */
int cr_profile_stats(cr_ctx* ctx, int64_t* out4);

/* ---- ordering front end ------------------------------------------------------------- */
/* OrderFormer.model forward -- models/model.py:206-233, called from predict :458-461.
 * boxes [B][L][4] bf16 (L <= 64; the reference pads every page to max_nums = 50 rows) -> scores [B][L] fp32
 * (= the bf16 decoder output).  Weights: the reference's `Transformer` state_dict keys under "orderformer."
 * (embedding.*, transformer_encoder.layers.N.*, decoder.*), loaded with cr_load_weight. */
int cr_orderformer(cr_ctx* ctx, const void* boxes, int B, int L, float* scores, void* stream);

/* ---- single operators (unit-parity tests and profiling) --------------------------------------- */
/* C = epi(A[M,K] . W[N,K]^T); epi: 0 store, 1 gelu, 2 layerscale+residual, 3 residual, 4 swiglu, 5 patch, 6 f32,
 * 7 decode partial sums (M <= 64, no bias: C = fp32 [S][M][N], S <= 8 K-slices summed by the consumer),
 * 8 row arg-max partials (cosine VQ: C = uint64 [M][ldc], one per row and 64-column block: bits of the bf16-rounded maximum in the
 *   high word, its first column in the low word; M > 64 tile kernels only).
 * Bits 8..15 of epi pin a kernel for the unit tests: 0 dispatcher's choice, 1 128x128 tiles, 2 256x256 persistent,
 * 3 weight-streaming (M <= 64), 5 / 6 the 256x256 kernel with its 16- / 32-MFMA-slot schedule (the same sums: the same bits);
 * a pinned kernel that cannot take the shape returns CR_ERR_ARG.
 * Bits 18-19 (M <= 64, bf16 weights, the weight-streaming kernels): W is cr_op_decode_swizzle's copy of the weight (ldw ignored) -- 1: plain 16-row
 * tiles (its which = 1..4), 2: wqkv's RoPE tile order (which = 0; epi 7 only: the K-slices still land in their nn.Linear columns).  The same bits as
 * from the nn.Linear layout. */
int cr_op_gemm(int epi, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
               const void* bias, const void* scale, const void* res, int64_t ldr, int M, int N, int K, int group,
               void* stream);
/* With bit 16 of epi set, W is e4m3 bytes [N][ldw] and `scale` is float[N] (one per row): C = epi((A . W^T) * scale); M <= 64. */
/* With bits 16 AND 17 set, A is e4m3 bytes [M][lda] too and `res` is float[M] (one scale per row of A):
 * C = epi((A . W^T) * res[m] * scale[n] + bias); the 256x256 kernel's e4m3 instance: K % 256 == 0, N % 64 == 0, epi 0 | 1 | 4 | 6. */
/* rows of a bf16 matrix -> e4m3 bytes q [N][K] + scale [N] (max|w| / 448 per row), as cr_enable_fp8_decode builds them */
int cr_op_quantize_fp8(const void* w, int64_t ldw, int N, int K, void* q, float* scale, void* stream);
int cr_op_layernorm(const void* in, void* out, const void* gamma, const void* beta, int64_t rows, int n, float eps,
                    int pixel_shuffle, void* stream);
int cr_op_rmsnorm(const void* in, void* out, const void* gamma, int64_t rows, int n, float eps, void* stream);
/* the same norms with the fp8 output cr_enable_fp8_mfma uses: beta == NULL selects RMSNorm (n = 4096), otherwise LayerNorm
 * (n = 1024 | 4096); out8 = e4m3 [rows][n], out_scale = float[rows] */
int cr_op_norm_fp8(const void* in, const void* gamma, const void* beta, int64_t rows, int n, float eps, void* out8, float* out_scale,
                   float* next_scale, const float* next_bound, void* stream);
/* next_scale / next_bound (both or neither): next_scale[row] = (1.13 * ||y_row||_2 * next_bound[0] + next_bound[1]) / 448 -- an upper bound
 * of what the next linear (largest weight-row norm next_bound[0], largest |bias| next_bound[1], device floats) can make of this row,
 * used as the e4m3 scale of THAT linear's output row by cr_op_gemm_q8 / the ViT's fc1 under cr_enable_fp8_mfma. */
/* C8 = e4m3(bf16(gelu(bf16((A8 . W8^T) * ascale[m] * wscale[n] + bias))) / c8scale[m]), bytes [M][N]: fc1's output as fc2's operand */
int cr_op_gemm_q8(const void* a8, const float* ascale, const void* w8, const float* wscale, const void* bias, const float* c8scale,
                  void* c8, int M, int N, int K, void* stream);
/* One small-batch decode GEMM with its neighbours folded in (gemm_decode.hip; M <= 8 rows, InternLM2.5-7B shapes): which = 0 wqkv
 * [RMSNorm(xres) -> GEMM -> RoPE + split: q_out rows, K / V rows into kc / vc at (seqs[m], lens[seqs[m]])], 1 wo [X -> GEMM -> xio += ],
 * 2 w1|w3 [RMSNorm(xres) -> GEMM -> SwiGLU -> C], 3 w2 [as 1], 4 LM head [RMSNorm(xres) -> GEMM -> fp32 C].  Replaces, per decode step of
 * InternLM2DecoderLayer.forward (modeling_internlm2.py:621-681): the norm / linear / rotary / residual statements around each linear.
 * Test and tuning entry point (cr_llm_decode drives the same launcher).  flags: bit 8 (256) = W is in the decode layout cr_op_decode_swizzle
 * writes (ldw ignored); other bits 0. */
int cr_op_decode_gemm(int which, int flags, const void* W, int64_t ldw, int M, int N, int K, const void* X, int64_t ldx, const void* xres,
                      const void* gamma, float eps, void* xio, void* C, int64_t ldc, const void* cosT, const void* sinT, void* q_out, void* kc,
                      void* vc, const int32_t* seqs, const int32_t* lens, int max_tokens, void* stream);
/* The decode layout of a weight for cr_op_decode_gemm's `which`: one 1 KiB block per (16-row tile, 32-deep k-step), lane l's eight elements at l * 16
 * bytes, in the kernel's own tile order (which = 0: RoPE tiles of 8 + 8 rows) -- a wave's load is one contiguous KiB instead of 16 rows x 64 bytes.
 * dst: ceil(N / 16) * 16 * K bf16.  cr_finalize keeps such a copy of every LLM linear (+15 GB on InternLM2.5-7B) for decode batches of <= 8 rows. */
int cr_op_decode_swizzle(int which, const void* W, int64_t ldw, int N, int K, void* dst, void* stream);
/* q/k/v/o addressed as base + b*bs + row*rs + head*hs (elements) */
int cr_op_attention(const void* q, const void* k, const void* v, void* o, const int64_t* strides12, int B, int H,
                    int Sq, int Sk, int head_dim, int kv_group, int causal, int q_pos0, float q_prescale, float s_div,
                    void* stream);

/* Batched decode attention alone (test / tuning entry; cr_llm_decode drives the same launcher): row b = one new token of cache slot seqs[b], its H_kv x 4 query heads x 128 at
 * q + b * 4096 (InternLM2's GQA layout: the 4 query heads of KV head h at (4 h .. 4 h + 3) * 128), keys / values [n_slots][8][max_tokens][128] bf16, lens[slot] + 1 keys per row
 * (the new token's K / V row is already in the cache), scores bf16 -> / s_div -> bf16, fp32 softmax, bf16 probabilities (modeling_internlm2.py:393-410); splits of 256 keys, partials in
 * `scratch` (cr_op_decode_attention_scratch_floats(B, max_keys) floats), merged in index order.  which = 0: the dispatcher's choice (attention_decode.hip), 1: the matrix-core split
 * kernel of attention.hip.  seqs / lens are DEVICE arrays.  ABI v10. */
int64_t cr_op_decode_attention_scratch_floats(int B, int max_keys);
int cr_op_decode_attention(int which, const void* q, const void* kc, const void* vc, int max_tokens, const int32_t* seqs, const int32_t* lens, int B, int max_keys,
                           float s_div, float* scratch, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CALLIREADER_HIP_H */
