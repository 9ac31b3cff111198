#pragma once
#include "common.hpp"

// Strides are in elements.  Q/K/V/O are addressed as base + b*bs + row*rs + head*hs + d.
struct AttnParams {
    const bf16* Q; const bf16* K; const bf16* V; bf16* O;
    int64_t q_bs, q_rs, q_hs;
    int64_t k_bs, k_rs, k_hs;
    int64_t v_bs, v_rs, v_hs;
    int64_t o_bs, o_rs, o_hs;
    int B, H, Sq, Sk;
    int kv_group;        // query heads per kv head (GQA): kv head = head / kv_group
    int q_pos0;          // causal: absolute position of query row 0 (keys start at position 0)
    float q_prescale;    // multiply q in bf16 before Q.K^T (ViT: 64^-0.5, exact); 1 = off
    float s_div;         // divide bf16-rounded scores (LLM: sqrt(128)); 1 = off
    // batched decode over a KV cache: batch b reads cache slot seq_map[b] (or b) holding sk_arr[slot] + sk_add keys
    const int32_t* seq_map;
    const int32_t* sk_arr;
    int sk_add;
    // split-KV (decode): workgroup x = split of SPLIT_TILES key tiles; partial (m, l) and un-normalised O go to
    // part_ml [B][H][nsplit][Sq][2] / part_o [B][H][nsplit][Sq][D] (fp32), merged by launch_attn_combine
    int nsplit;
    float* part_ml;
    float* part_o;
    // causal prefill of several pages in ONE launch: batch b is segment seg[4b..4b+3] = {first row, rows, position of the first
    // row, cache slot}: Q / O rows start at `first row` (q_bs / o_bs unused), keys = position + rows, Sq = the longest segment
    const int32_t* seg;
    bool force_matrix_core;   // decode: keep the matrix-core split kernel even where attention_decode.hip's streaming kernel qualifies (cr_op_decode_attention which = 1: A/B)
    // decode with RoPE + split FOLDED IN (attention_decode.hip, 9..64-row batches): instead of reading Q and finding the new token's K / V row in the cache, the
    // kernel sums wqkv's K-slice partial sums [qkv_splits][B][qkv_ld] fp32 itself (slice order, rounded once), rotates q and k (rope_split_kernel's arithmetic) and
    // the workgroup whose split holds the new position writes the K / V row to the cache.  Row layout of qkv: [H groups][4 q | k | v][128].
    const float* qkv_part;
    int qkv_splits;
    int64_t qkv_ld;
    const bf16* rope_cos;     // [max_pos][128]
    const bf16* rope_sin;
};

bool decode_attn_fold_supported(const AttnParams& p, int head_dim);

constexpr int ATTN_SPLIT_TILES = 4;      // 256 keys per split: depends only on the row's own key count

int launch_flash_attn(const AttnParams& p, int head_dim, bool causal, hipStream_t stream);
// decode: Sq <= 32 query rows per (batch, head), non-causal over each sequence's own keys, split over the keys
int launch_flash_attn_split(const AttnParams& p, int head_dim, hipStream_t stream);
size_t attn_split_ws_floats(int B, int H, int Sq, int nsplit, int head_dim);
// attention_decode.hip: the split half of launch_flash_attn_split as a streaming vector-pipe kernel (d = 128, 4 query rows per (batch, head): InternLM2's GQA
// group); launch_flash_attn_split takes it by itself when the shape qualifies (CR_DECODE_ATTN=0: the matrix-core split kernel)
bool decode_attn_supported(const AttnParams& p, int head_dim);
int launch_decode_attn(const AttnParams& p, hipStream_t stream);

// ViT layout (S = 1 + 128 n tokens, d = 64, no mask): attention_vit.hip.  launch_flash_attn takes this path by itself when
// p.part_ml points at vit_attn_ws_floats(B, H, S) floats of scratch (the CLS query's partials) and the shape qualifies.
bool vit_attn_supported(const AttnParams& p, int head_dim, bool causal);
size_t vit_attn_ws_floats(int B, int H, int S);
int launch_vit_attn(const AttnParams& p, hipStream_t stream);
