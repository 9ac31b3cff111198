// Context of libcallireader_hip.so: weights (library-owned device copies), derived
// tensors, a grow-only device workspace.  Host-side C++ only; no torch types.
#pragma once
#include <atomic>
#include <deque>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/callireader_hip.h"
#include "common.hpp"

struct DevTensor {
    void* ptr = nullptr;
    size_t bytes = 0;
    int dtype = CR_BF16;
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct cr_ctx {
    int device = 0;
    cr_model_desc d{};
    bool finalized = false;
    int prof_mode = 0;                  // cr_profile: 1 = every GEMM launch, 2 = only the tiled class (M >= 1024): leaves decode free to run as a graph
    int decode_graph = 0;               // CR_DECODE_GRAPH=1 replays the decode step as a captured hipGraph (measured slower here, see llm.hip); reset to 0 after a capture failure
    hipStream_t side = nullptr;         // blocking stream the decode graph is captured into and replayed on when the caller's stream is the
                                        // null stream (which cannot be captured); implicitly ordered with it
    uint64_t weight_gen = 0;            // bumped by cr_load_weight / cr_finalize: captured graphs hold weight pointers
    bool fp8_decode = false;            // cr_enable_fp8_decode: batched decode streams e4m3 copies of the LLM's linear weights (half the bytes)
    int fp8_mfma = 0;                   // cr_enable_fp8_mfma level: 1 = the norm-fed linears of the ViT, the projector and the LLM prefill run e4m3 x e4m3 on the
                                        // matrix cores; 2 = also the linears whose input no norm produces (ViT fc2 via fc1's e4m3 epilogue, LLM wo / w2 via a quantiser pass)
    bf16* probe_dst = nullptr;          // cr_llm_hidden_probe: [layers + 1][probe_rows][4096] rows of the residual stream of the next prefills
    int probe_row0 = 0, probe_rows = 0;
    bool fused_decode = true;           // decode batches of <= DECODE_FUSED_MAX_ROWS (8) rows take gemm_decode.hip (six launches per layer, the same bits); CR_DECODE_FUSED=0: the separate kernels
    bool prefill_last_rows = true;      // prefill: the final decoder layer's attention / wo / w1|w3 / w2 on each page's last row only (bit-identical; CR_PREFILL_LAST_ROWS=0: all rows)
    bool no_sliced_decode = false;      // CR_NO_SLICED_DECODE=1: decode through the one-tile weight-streaming GEMMs (A/B aid)
    bool perceiver_v1 = false;          // CR_PERCEIVER_ATTN_V1=1 at cr_create: round 1's perceiver attention kernel (the bit-for-bit yardstick of the current one)
    bool fold_rope = true;              // 9..64-row decode: RoPE + split + the new token's cache row inside the attention kernel (attention_decode.hip FOLD; bit-identical); CR_DECODE_FOLD_ROPE=0 at cr_create: rope_split_kernel as its own launch
    std::unordered_map<std::string, DevTensor> w;
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bool borrowed = false;              // cr_share_weights: the tensors in `w` belong to another context (not freed here)
    // weight_gen mirrored into a cell that OUTLIVES the context (cr_destroy stores ~0 into it): a borrower keeps the cell of the context that
    // really owns the tensors (the root of a chain of cr_share_weights calls), never a pointer to the context itself
    std::shared_ptr<std::atomic<uint64_t>> gen_cell = std::make_shared<std::atomic<uint64_t>>(0);
    std::shared_ptr<std::atomic<uint64_t>> owner_cell;      // borrowed: the owning context's cell, and its value when the map was copied: a reload /
    uint64_t owner_gen = 0;             //     re-finalize / fp8 toggle / destroy of the owner frees or replaces tensors this copy still points at -> the
                                        //     stage entry points refuse to run (ctx_share_ok)
    // small persistent device scratch (counters, argmax partials)
    char* scratch = nullptr;
    size_t scratch_bytes = 0;
    // measurement (cr_profile): event pairs around GEMM launches
    bool prof = false;
    struct ProfRec { hipEvent_t a, b; double flops, bytes; int big; };
    std::deque<ProfRec> prof_recs;      // launches whose events have not been read back yet (bounded: see ctx_gemm)
    std::vector<hipEvent_t> prof_pool;
    double prof_acc[2][4] = {};         // [tiled M >= 1024 | the rest][launches, ms, flops, bytes] of the retired records
    int64_t prof_issued = 0, prof_retired = 0, prof_lost = 0, prof_peak_pending = 0;
};

inline void cr_bump_gen(cr_ctx* c) { c->weight_gen++; c->gen_cell->store(c->weight_gen, std::memory_order_release); }

// e4m3 copy + per-row fp32 scale of weight `name` as "fp8.<name>" / "fp8s.<name>" (llm.hip); a no-op when already built for this shape
int build_fp8_copy(cr_ctx* c, const std::string& name, int k_multiple, hipStream_t st);
// C = epi((A8 . W8^T) * ascale[m] * wscale[n] + bias): both operands e4m3 (gemm256's F8 instance)
int ctx_gemm_f8(cr_ctx* c, int epi, const void* a8, const float* ascale, const DevTensor* w8, const DevTensor* ws, void* C, int64_t ldc,
                const bf16* bias, int M, int N, int K, hipStream_t st, const bf16* res = nullptr, int64_t ldr = 0);

// CR_OK, or CR_ERR_STATE when this context borrows weights whose owner has changed them since cr_share_weights
int ctx_share_ok(const cr_ctx* c, const char* who);

// GEMM launch used by every stage: validates, launches, and (when profiling) brackets the launch with events.
int ctx_gemm(cr_ctx* c, int epi, const GemmParams& p, hipStream_t st);

void cr_set_error(const char* fmt, ...);
int cr_fail(int code, const char* fmt, ...);
#define CR_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) return cr_fail(CR_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define CR_TRY(call)                       \
    do {                                   \
        int r_ = (call);                   \
        if (r_ != CR_OK) return r_;        \
    } while (0)

// weight lookup helpers (api.hip)
const bf16* W(cr_ctx* c, const std::string& name);
const DevTensor* WT(cr_ctx* c, const std::string& name);
int ws_ensure(cr_ctx* c, size_t bytes);

// simple bump allocator over the workspace
struct Arena {
    char* base; size_t off = 0;
    explicit Arena(char* b) : base(b) {}
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
};
