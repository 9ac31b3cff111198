// Shared device/host helpers for the CalliReader gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CR_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define CR_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

#include "../../include/callireader_hip.h"   // error codes of the C ABI

// ---- per-DEVICE one-time launch set-up (host) -------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize and the CU count belong to a device, not to the process: a launcher keeps
// one bit per device (the calling thread's current device) in a function-local atomic; setting the attribute twice
// from two threads is harmless, so no lock is needed.  Returns false when the HIP call fails.
inline bool cr_dyn_lds_once(std::atomic<uint64_t>& done, const void* fn, int lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}
inline int cr_device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = cus[dev & 63].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("CR_CUS")) { const int v = atoi(e); if (v > 0 && v <= n) n = v; }     // tuning aid: persistent grids for a CU-masked stream
    cus[dev & 63].store(n, std::memory_order_relaxed);
    return n;
}

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }      // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
// Round to bf16 and back (RNE).  Integer arithmetic on purpose: written as (float)(bf16)x, LLVM narrows
// fptrunc(op(fpext a, fpext b)) chains into bf16 ops and the AMDGPU backend then evaluates them in fp32 WITHOUT the
// intermediate rounding (seen as a single v_fma_f32), which silently drops the reference's rounding points.
// Inf stays Inf; NaN is not preserved (never needed at these call sites).
__device__ __forceinline__ float rbf(float x) {
    unsigned u = __float_as_uint(x);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return __uint_as_float(u);
}

// (a, b) -> bf16-rounded (RNE) values as floats.  The VECTOR conversion is what selects one v_cvt_pk_bf16_f32 for the
// pair (two scalar casts compile to two conversions plus shifts); unpacking through integer ops keeps LLVM from
// folding the rounding away.  Deliberately NOT inline asm: as the first reader of MFMA results an asm statement
// gets no MFMA->VALU wait states from hipcc (NaNs observed).
__device__ __forceinline__ void round_pair_bf16(float a, float b, float& ra, float& rb2) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 v2 = {a, b};
    const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));
    ra = __uint_as_float(pk << 16);
    rb2 = __uint_as_float(pk & 0xffff0000u);
}

// GELU (erf form) as the reference computes it in fp32: (x * 0.5) * (1 + erf(x / sqrt 2)).
// erf is evaluated branch-free as 1 - erfc(|t|), erfc(t) = exp(-t^2) * sum_{k=1..6} a_k u^k, u = 1 / (1 + p t): the
// Abramowitz-Stegun 7.1.26 form re-fitted with a sixth term (max |error| 1.1e-8 in exact arithmetic, ~1e-7 as
// evaluated in fp32: the same class as any fp32 erff; the negative tail reproduces the reference's own
// 1 + erf cancellation, including -0 below x ~ -5.6).  ~21 VALU issue slots against ~50 for the two divergent
// branches of the device library's erff, which made the fc1 epilogue 24 % of that GEMM.  Checked on all finite bf16
// inputs against torch's CPU GELU in tests/test_gpu_ops.py (the input of this function is always a bf16 value).
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = fabsf(x) * 0.70710678118654752440f;
    const float u = __builtin_amdgcn_rcpf(fmaf(0.29046997f, t, 1.0f));
    float s = 1.3864057f;
    s = fmaf(s, u, -2.8561068f);
    s = fmaf(s, u, 3.3568153f);
    s = fmaf(s, u, -1.7148181f);
    s = fmaf(s, u, 0.73626f);
    s = fmaf(s, u, 0.0914439f);
    s *= u;
    const float e = __builtin_amdgcn_exp2f(t * (t * -1.4426950408889634f));
    const float erf_abs = 1.0f - s * e;
    return (x * 0.5f) * (1.0f + copysignf(erf_abs, x));
}
// The same function on a PAIR: every multiply / FMA / add is one packed instruction (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32: IEEE
// results identical to the scalar forms, so each component is bit for bit gelu_erf of its input; rcp and exp2 have no packed form).
// 17 of the 21 issue slots halve: worth a quarter of the vector work of the fc1 / mlp1 epilogues (scripts/ubench/valu_issue.hip: a
// packed fp32 op issues in ~1.4x the slot of a plain one and does two).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 t = ax * 0.70710678118654752440f;
    const f32x2 d = __builtin_elementwise_fma(f32x2{0.29046997f, 0.29046997f}, t, f32x2{1.0f, 1.0f});
    const f32x2 u = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 s = {1.3864057f, 1.3864057f};
    s = __builtin_elementwise_fma(s, u, f32x2{-2.8561068f, -2.8561068f});
    s = __builtin_elementwise_fma(s, u, f32x2{3.3568153f, 3.3568153f});
    s = __builtin_elementwise_fma(s, u, f32x2{-1.7148181f, -1.7148181f});
    s = __builtin_elementwise_fma(s, u, f32x2{0.73626f, 0.73626f});
    s = __builtin_elementwise_fma(s, u, f32x2{0.0914439f, 0.0914439f});
    s = s * u;
    const f32x2 a = t * (t * -1.4426950408889634f);
    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 erf_abs = f32x2{1.0f, 1.0f} - s * e;
    const f32x2 sg = {copysignf(erf_abs[0], x[0]), copysignf(erf_abs[1], x[1])};
    return (x * 0.5f) * (f32x2{1.0f, 1.0f} + sg);
}
// 16 fp32 values times `inv` -> 16 e4m3 bytes (OCP e4m3fn, round to nearest even, saturating): one 16-byte store
__device__ __forceinline__ void store16_e4m3(unsigned char* dst, const float* y, float inv) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
    u32x4_ w = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned d = 0;
        d = __builtin_amdgcn_cvt_pk_fp8_f32(y[4 * q] * inv, y[4 * q + 1] * inv, d, false);
        d = __builtin_amdgcn_cvt_pk_fp8_f32(y[4 * q + 2] * inv, y[4 * q + 3] * inv, d, true);
        w[q] = d;
    }
    *(u32x4_*)dst = w;
}

__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- GEMM -----------------------------------------------------------------
// C[M,N] = epilogue(A[M,K] . W[N,K]^T); A and W row-major with K contiguous
// (exactly how nn.Linear stores its weight), K % 64 == 0.
enum GemmEpi {
    EPI_STORE = 0,    // bf16(acc + bias)
    EPI_GELU = 1,     // bf16(gelu(bf16(acc + bias)))
    EPI_LS_RES = 2,   // bf16(res + bf16(bf16(acc + bias) * scale))         (ViT LayerScale + residual)
    EPI_RES = 3,      // bf16(res + bf16(acc + bias))                       (LLM / resampler residual)
    EPI_SWIGLU = 4,   // bf16(bf16(silu(bf16(g))) * bf16(u)), W rows interleaved [8 gate | 8 up]
    EPI_PATCH = 5,    // patch-embed: bf16(bf16(acc + bias) + pos[1 + m % G]) -> row (m / G) * (G + 1) + 1 + m % G
    EPI_F32 = 6,      // float(bf16(acc + bias))                            (logits: bf16 GEMM then .float())
    EPI_PARTIAL = 7,  // decode only (M <= 64): fp32 partial sums [S][M][N] of S K-slices, no bias; the consumer kernel sums them
    EPI_GELU_Q8 = 9,  // e4m3 instance only: e4m3(bf16(gelu(bf16(acc + bias))) / c8scale[m]) -- fc1's output as fc2's fp8 operand, no bf16 copy
    EPI_ARGMAX = 8,   // cosine VQ: C[m][b] = {col, bits(max)} of bf16(acc) over the 64 columns of block b (first max wins); nothing else stored
};

struct GemmParams {
    const bf16* A; int64_t lda;
    const bf16* W; int64_t ldw;
    void* C; int64_t ldc;
    const bf16* bias;          // [N] or nullptr
    const bf16* scale;         // [N] (EPI_LS_RES)
    const bf16* res; int64_t ldr;   // residual rows (EPI_LS_RES / EPI_RES) or pos-emb (EPI_PATCH)
    int M, N, K;
    int group;                 // EPI_PATCH: patches per tile (1024)
    int kernel;                // 0 = dispatcher's choice; 128 | 256 | 1 (skinny) pin one (cr_op_gemm's tests only)
    int slots;                 // gemm256 only: 0 = schedule picked per shape; 16 | 32 pin one of its two schedules (tests: the same bits)
    // fp8 weight streaming (decode, M <= 64 only): W points at e4m3 bytes [N][ldw], wscale[n] restores row n (C = (X . W8^T) * wscale)
    int w8;
    const float* wscale;
    // fp8 matrix-core path (gemm256 only, M large): A points at e4m3 bytes [M][lda] as well (w8 set too), K counts fp8 elements
    // (K % 256 == 0), ascale[m] restores row m: C = epi((A8 . W8^T) * ascale[m] * wscale[n] + bias)
    int a8;
    const float* ascale;
    const float* c8scale;      // EPI_GELU_Q8: C is e4m3 bytes [M][ldc], row m divided by c8scale[m] (an upper bound of the row's magnitude / 448)
    // decode layout (M <= 64, bf16 weights, the weight-streaming kernels only): W points at decode_swizzle_weight's copy -- one contiguous KiB per 16-row tile
    // and 32-deep k-step, lane l's 16 bytes at l * 16 (ldw unused).  1: tiles are rows 16 t .. 16 t + 15; 2: wqkv's RoPE tile order (8 rows of a head's
    // first half + the 8 rows 64 further on; EPI_PARTIAL only: the sums land in their nn.Linear columns).  N % 16 == 0 except for layout 1's last tile.
    int wsw;
};
// row of W that tile t, row r of the RoPE tile order holds (gemm_decode.hip: wrow; head dim 128)
__host__ __device__ __forceinline__ int rope_tile_row(int t, int r) { return (t >> 3) * 128 + (r < 8 ? 8 * (t & 7) + r : 64 + 8 * (t & 7) + (r - 8)); }

// EPI_ARGMAX partial of one row and one 64-column block: the bf16-rounded maximum (as fp32 bits, high word) and its column
// (low word); -inf / column 0x7fffffff for a block with no valid column.  Merging takes the larger value, then the
// smaller column: torch.max's first-index rule (models/similarity.py:21).
__device__ __forceinline__ void argmax_merge(float& bv, int& bc, float v, int c) {
    if (v > bv || (v == bv && c < bc)) { bv = v; bc = c; }
}
__device__ __forceinline__ unsigned long long argmax_pack(float v, int c) {
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)c;
}

int launch_gemm(int epi, const GemmParams& p, hipStream_t stream);
// K-slices of the EPI_PARTIAL decode GEMM for an [N, K] weight (depends on N and K only, never on M); 0 = unsupported
int gemm_partial_splits(int N, int K);
