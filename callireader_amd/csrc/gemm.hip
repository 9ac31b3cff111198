// bf16 MFMA GEMM for gfx950 with fused epilogues:  C = epi(A[M,K] . W[N,K]^T)
//
// Every dense contraction of the hot path is this shape (nn.Linear keeps W as
// [N,K], K contiguous), see SURVEY.md 2a rows V1,V4,V6,V7,P2,R1,Q1,L2,L6,L7,L8.
//
// Structure (128x128x64 tile, 4 waves of 64 lanes, each wave a 64x64 sub-tile
// as 4x4 v_mfma_f32_16x16x32_bf16 accumulators):
//   * global -> LDS by global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip),
//     two LDS buffers, the load of K-tile t+1 is issued before the MFMAs of tile t;
//   * LDS image is [row][64 bf16] = 128-B rows; the 16-B chunk index is XORed
//     with (row>>1)&7.  LDS-DMA writes lane-linearly, so the XOR is applied to
//     the per-lane SOURCE address and again on the ds_read_b128 (conflict-free
//     for the 16x16x32 operand fragment: lanes 0-15 rows r, chunk c);
//   * workgroup id -> tile: bijective XCD remap (each XCD's L2 gets a contiguous
//     chunk of tiles) then 8-row super-groups so 64 co-resident tiles share
//     8 A-panels and 8 W-panels;
//   * epilogue: accumulators -> per-wave LDS stage (fp32) -> each lane owns 8
//     contiguous columns of a row: 16-B loads of bias/scale/residual, the
//     reference's bf16 rounding sequence, one 16-B store (full 128-B lines).
//   * rows beyond M / N are clamped on load and masked on store.
#include <stdlib.h>

#include "gemm_epilogue.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_MAIN = 2 * (BM + BN) * BK * 2;     // 65536
constexpr int LDS_STAGE = 4 * 64 * STAGE_LD * 4;     // 69632
constexpr int LDS_BYTES = LDS_STAGE > LDS_MAIN ? LDS_STAGE : LDS_MAIN;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- workgroup -> tile (XCD-aware, bijective for any grid size) ----
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    constexpr int GM = 8;
    const int per_group = GM * ntn;
    const int grp = pid / per_group;
    const int first_m = grp * GM;
    const int gsz = min(ntm - first_m, GM);
    const int in_g = pid - grp * per_group;
    const int tile_m = first_m + in_g % gsz;
    const int tile_n = in_g / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane LDS-DMA source addresses (swizzle lives on the source side) ----
    const bf16* ga[4];
    const bf16* gb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rr = wave * 32 + i * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((rr >> 1) & 7);
        const int gm = min(m0 + rr, p.M - 1);
        const int gn = min(n0 + rr, p.N - 1);
        ga[i] = p.A + (int64_t)gm * p.lda + ch * 8;
        gb[i] = p.W + (int64_t)gn * p.ldw + ch * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 32768 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds(CR_GLB(ga[i] + (int64_t)kt * BK), CR_LDS(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CR_GLB(gb[i] + (int64_t)kt * BK), CR_LDS(base + 16384 + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int swz = (lane >> 1) & 7;
    const int a_off = (wm * 64 + (lane & 15)) * 128;
    const int b_off = 16384 + (wn * 64 + (lane & 15)) * 128;
    const int nk = p.K / BK;

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sbuf = smem + cur * 32768;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int ch = ((4 * s + (lane >> 4)) ^ swz) * 16;
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = *(const bf16x8*)(sbuf + a_off + i * 2048 + ch);
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = *(const bf16x8*)(sbuf + b_off + j * 2048 + ch);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue through a per-wave fp32 stage ----
    float* st = (float*)smem + wave * (64 * STAGE_LD);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                st[(i * 16 + (lane >> 4) * 4 + e) * STAGE_LD + j * 16 + (lane & 15)] = acc[i][j][e];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the stage is wave-private
    __builtin_amdgcn_wave_barrier();

    epilogue_subtile<EPI>(p, st, m0 + wm * 64, n0 + wn * 64, lane);
}

template <int EPI>
int launch_t(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)gemm128_kernel<EPI>, LDS_BYTES)) return CR_ERR_HIP;
    hipLaunchKernelGGL(gemm128_kernel<EPI>, dim3(ntm * ntn), dim3(256), LDS_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

}  // namespace

bool gemm256_supported(int epi, const GemmParams& p);
int launch_gemm256(int epi, const GemmParams& p, hipStream_t stream);
int launch_gemm256_f8(int epi, const GemmParams& p, hipStream_t stream);
bool gemm_skinny_supported(int epi, const GemmParams& p);
int launch_gemm_skinny(int epi, const GemmParams& p, hipStream_t stream);
bool gemm_tail_supported(int epi, const GemmParams& p);
int launch_gemm_tail(int epi, const GemmParams& p, hipStream_t stream);

int launch_gemm(int epi, const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0) return CR_ERR_ARG;
    if ((p.lda & 7) || (p.ldw & 7)) return CR_ERR_ARG;
    if (p.wsw) {                                                 // decode-layout weights: the weight-streaming kernels only (M <= 64; bf16, or e4m3 in plain tile order)
        if ((p.w8 && p.wsw != 1) || p.a8 || p.wsw > 2 || (p.wsw == 2 && epi != EPI_PARTIAL) || ((uintptr_t)p.W & 15) || !gemm_skinny_supported(epi, p)) return CR_ERR_ARG;
        return launch_gemm_skinny(epi, p, stream);
    }
    if (p.a8) return launch_gemm256_f8(epi, p, stream);                                                  // e4m3 x e4m3 on the matrix cores
    if (p.w8) return gemm_skinny_supported(epi, p) ? launch_gemm_skinny(epi, p, stream) : CR_ERR_ARG;   // fp8 weights: decode kernel only
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.C & 15)) return CR_ERR_ARG;
    static const int env_force = [] { const char* e = getenv("CR_GEMM_FORCE"); return e ? atoi(e) : 0; }();   // tuning aid: 128 | 256
    const int force = p.kernel ? p.kernel : env_force;
    if (force == 1) return gemm_skinny_supported(epi, p) ? launch_gemm_skinny(epi, p, stream) : CR_ERR_ARG;
    if (force == 256) return (p.K % 128) == 0 ? launch_gemm256(epi, p, stream) : CR_ERR_ARG;
    if (force != 128) {
        if (gemm_skinny_supported(epi, p)) return launch_gemm_skinny(epi, p, stream);   // decode: stream W once from HBM
        if (gemm256_supported(epi, p)) {                                                // large M: persistent 256x256 kernel
            // A few rows past a multiple of 256 can cost a whole extra round of tiles (BASELINE config 2: 32 tiles = 128 x 256 + 32
            // rows; with N = 1024 that is 2 rounds + 4 tiles = 3 rounds).  When cutting them off saves a round, they go through a
            // one-wave-per-workgroup kernel (<= 64 rows, gemm_skinny.hip: launch_gemm_tail) or the 128x128 kernel instead, whose K order
            // and rounding are the same (a row's result does not depend on the kernel:
            // tests/test_gpu_full_depth.py chunking test, tests/test_gpu_ops.py::test_gemm_tail_rows_take_the_small_kernel).
            const int r = p.M % 256, cus = cr_device_cus();
            const long ntn = (p.N + 255) / 256, tm = p.M / 256;
            if (r > 0 && tm > 0 && epi != EPI_PATCH && epi != EPI_ARGMAX && (tm * ntn + cus - 1) / cus < ((tm + 1) * ntn + cus - 1) / cus &&
                (long)((r + 127) / 128) * ((p.N + 127) / 128) <= 2L * cus) {
                GemmParams a = p, b = p;
                a.M = p.M - r;
                b.M = r;
                b.A = p.A + (int64_t)a.M * p.lda;
                b.C = (char*)p.C + (int64_t)a.M * p.ldc * (epi == EPI_F32 ? 4 : 2);
                if (p.res) b.res = p.res + (int64_t)a.M * p.ldr;
                const int rc = launch_gemm256(epi, a, stream);
                if (rc != CR_OK) return rc;
                if (gemm_tail_supported(epi, b)) return launch_gemm_tail(epi, b, stream);
                b.kernel = 128;
                return launch_gemm(epi, b, stream);
            }
            return launch_gemm256(epi, p, stream);
        }
    }
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE>(p, stream);
        case EPI_GELU: return launch_t<EPI_GELU>(p, stream);
        case EPI_LS_RES: return (p.scale && p.res) ? launch_t<EPI_LS_RES>(p, stream) : CR_ERR_ARG;
        case EPI_RES: return p.res ? launch_t<EPI_RES>(p, stream) : CR_ERR_ARG;
        case EPI_SWIGLU: return (p.N % 16 == 0) ? launch_t<EPI_SWIGLU>(p, stream) : CR_ERR_ARG;
        case EPI_PATCH: return (p.res && p.group > 0) ? launch_t<EPI_PATCH>(p, stream) : CR_ERR_ARG;
        case EPI_F32: return launch_t<EPI_F32>(p, stream);
        case EPI_ARGMAX: return launch_t<EPI_ARGMAX>(p, stream);
    }
    return CR_ERR_ARG;
}
