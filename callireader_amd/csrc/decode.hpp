// Small-batch decode GEMMs with their neighbours folded in (gemm_decode.hip)
#pragma once
#include "common.hpp"

constexpr int DECODE_FUSED_MAX_ROWS = 8;     // beyond this the separate kernels are faster (gemm_decode.hip's header, profiles/round4/)

// K-slice counts the fused kernels reproduce as "virtual slices": they must be gemm_skinny.hip's partial_geom() picks for the same shapes, or a row's bits
// would differ between a <= 8-row batch (fused) and a larger one (K-sliced partials summed by the consumer); run_layers checks it before taking the fused path
constexpr int DEC_SLICES_WQKV = 2, DEC_SLICES_WO = 4, DEC_SLICES_W2 = 4;

enum DecodeGemm { DEC_WQKV = 0, DEC_WO = 1, DEC_W13 = 2, DEC_W2 = 3, DEC_HEAD = 4 };

struct DecodeGemmParams {
    const bf16* W; int64_t ldw;          // [N][K] as nn.Linear stores it (w1|w3: the interleaved derived tensor) -- or, with `swizzled`, its decode layout
    int swizzled;                        //   (decode_swizzle_weight: 1 KiB per 16-row tile and 32-deep k-step; ldw unused)
    int M, N, K;
    const bf16* X; int64_t ldx;          // DEC_WO / DEC_W2: activations [M][K]
    const bf16* xres;                    // DEC_WQKV / DEC_W13 / DEC_HEAD: residual rows [M][4096] the prologue normalises
    const bf16* gamma; float eps;        //   ... with this RMSNorm weight
    bf16* xio;                           // DEC_WO / DEC_W2: residual rows [M][4096], updated in place
    void* C; int64_t ldc;                // DEC_W13: bf16 [M][N / 2]; DEC_HEAD: float [M][N]
    // DEC_WQKV: RoPE tables [max_pos][128], q rows [M][4096], this layer's K / V cache [n_seqs][8][max_tokens][128], cache slot and
    // position (= tokens cached so far) of every row
    const bf16 *cosT, *sinT;
    bf16 *q_out, *kc, *vc;
    const int32_t *seqs, *lens;
    int max_tokens;
    int flags;                           // tuning knobs (cr_op_decode_gemm / CR_DEC_FLAGS): bit 0 forces, bit 1 forbids the one-wave-per-row RMSNorm prologue of w1|w3 / the LM head
};

bool decode_fused_supported(int M, int ff);
int launch_decode_gemm(int which, const DecodeGemmParams& p, hipStream_t stream);
// the decode layout of one of the five weights (DEC_WQKV bakes the RoPE tile order in): `dst` holds decode_swizzled_bytes(N, K)
size_t decode_swizzled_bytes(int N, int K);
int decode_swizzle_weight(int which, const bf16* W, int64_t ldw, int N, int K, bf16* dst, hipStream_t stream);
// the e4m3 copies of cr_enable_fp8_decode in their decode layout (1 KiB per 16-row tile and 64-deep k-step): gemm_skinny.hip's W8 instances with GemmParams::wsw = 1
size_t decode_swizzled8_bytes(int N, int K);
int decode_swizzle_weight8(const unsigned char* W, int64_t ldw, int N, int K, unsigned char* dst, hipStream_t stream);
