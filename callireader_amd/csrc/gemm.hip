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
#include "common.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_LD = 68;                         // fp32 words per staged row (64 + 4 pad)
constexpr int LDS_MAIN = 2 * (BM + BN) * BK * 2;     // 65536
constexpr int LDS_STAGE = 4 * 64 * STAGE_LD * 4;     // 69632
constexpr int LDS_BYTES = LDS_STAGE > LDS_MAIN ? LDS_STAGE : LDS_MAIN;

template <int EPI>
__device__ __forceinline__ void epilogue_row8(const GemmParams& p, int gm, int gn, const float* v) {
    // v[0..7]: fp32 accumulators of row gm, columns gn..gn+7 (gn % 8 == 0, gn + 8 <= N)
    float x[8];
    if (p.bias) {
        bf16x8 b = *(const bf16x8*)(p.bias + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = rbf(v[e] + bf2f(b[e]));
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = rbf(v[e]);
    }
    if (EPI == EPI_F32) {
        float* c = (float*)p.C + (int64_t)gm * p.ldc + gn;
        if ((p.ldc & 3) == 0) {
            *(f32x4*)c = f32x4{x[0], x[1], x[2], x[3]};
            *(f32x4*)(c + 4) = f32x4{x[4], x[5], x[6], x[7]};
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) c[e] = x[e];
        }
        return;
    }
    int64_t orow = gm;
    if (EPI == EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = gelu_erf(x[e]);
    } else if (EPI == EPI_LS_RES) {
        bf16x8 s = *(const bf16x8*)(p.scale + gn);
        bf16x8 r = *(const bf16x8*)(p.res + (int64_t)gm * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = bf2f(r[e]) + rbf(x[e] * bf2f(s[e]));
    } else if (EPI == EPI_RES) {
        bf16x8 r = *(const bf16x8*)(p.res + (int64_t)gm * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = bf2f(r[e]) + x[e];
    } else if (EPI == EPI_PATCH) {
        int t = gm / p.group, pi = gm - t * p.group;
        orow = (int64_t)t * (p.group + 1) + 1 + pi;
        bf16x8 r = *(const bf16x8*)(p.res + (int64_t)(1 + pi) * p.ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = x[e] + bf2f(r[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(x[e]);
    *(bf16x8*)((bf16*)p.C + orow * p.ldc + gn) = o;
}

template <int EPI>
__device__ __forceinline__ void epilogue_scalar(const GemmParams& p, int gm, int gn, float v) {
    float x = rbf(v + (p.bias ? bf2f(p.bias[gn]) : 0.0f));
    if (EPI == EPI_F32) { ((float*)p.C)[(int64_t)gm * p.ldc + gn] = x; return; }
    int64_t orow = gm;
    if (EPI == EPI_GELU) x = gelu_erf(x);
    else if (EPI == EPI_LS_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn]) + rbf(x * bf2f(p.scale[gn]));
    else if (EPI == EPI_RES) x = bf2f(p.res[(int64_t)gm * p.ldr + gn]) + x;
    else if (EPI == EPI_PATCH) {
        int t = gm / p.group, pi = gm - t * p.group;
        orow = (int64_t)t * (p.group + 1) + 1 + pi;
        x = x + bf2f(p.res[(int64_t)(1 + pi) * p.ldr + gn]);
    }
    ((bf16*)p.C)[orow * p.ldc + gn] = f2bf(x);
}

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

    if (EPI == EPI_SWIGLU) {
        // staged columns: [8 gate | 8 up] x 4 per 64-wide row -> 32 outputs per row
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 16 + (lane >> 2);
            const int oc = lane & 3;
            const int gm = m0 + wm * 64 + row;
            const int gno = (n0 + wn * 64) / 2 + oc * 8;
            if (gm < p.M && gno + 8 <= p.N / 2) {
                const float* sp = st + row * STAGE_LD + oc * 16;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float g = rbf(sp[e]), u = rbf(sp[8 + e]);
                    o[e] = f2bf(rbf(silu(g)) * u);
                }
                *(bf16x8*)((bf16*)p.C + (int64_t)gm * p.ldc + gno) = o;
            }
        }
        return;
    }
    const bool vec_ok = ((p.ldc & 7) == 0) || (EPI == EPI_F32);
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int row = it * 8 + (lane >> 3);
        const int c8 = (lane & 7) * 8;
        const int gm = m0 + wm * 64 + row;
        const int gn = n0 + wn * 64 + c8;
        if (gm >= p.M || gn >= p.N) continue;
        const float* sp = st + row * STAGE_LD + c8;
        if (gn + 8 <= p.N && vec_ok) {
            f32x4 v0 = *(const f32x4*)sp, v1 = *(const f32x4*)(sp + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            epilogue_row8<EPI>(p, gm, gn, v);
        } else {
            for (int e = 0; e < 8 && gn + e < p.N; e++) epilogue_scalar<EPI>(p, gm, gn + e, sp[e]);
        }
    }
}

template <int EPI>
int launch_t(const GemmParams& p, hipStream_t stream) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm128_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return CR_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm128_kernel<EPI>, dim3(ntm * ntn), dim3(256), LDS_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

}  // namespace

bool gemm_skinny_supported(int epi, const GemmParams& p);
int launch_gemm_skinny(int epi, const GemmParams& p, hipStream_t stream);

int launch_gemm(int epi, const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0) return CR_ERR_ARG;
    if ((p.lda & 7) || (p.ldw & 7)) return CR_ERR_ARG;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.C & 15)) return CR_ERR_ARG;
    if (gemm_skinny_supported(epi, p)) return launch_gemm_skinny(epi, p, stream);   // decode: stream W once from HBM
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE>(p, stream);
        case EPI_GELU: return launch_t<EPI_GELU>(p, stream);
        case EPI_LS_RES: return (p.scale && p.res) ? launch_t<EPI_LS_RES>(p, stream) : CR_ERR_ARG;
        case EPI_RES: return p.res ? launch_t<EPI_RES>(p, stream) : CR_ERR_ARG;
        case EPI_SWIGLU: return (p.N % 16 == 0) ? launch_t<EPI_SWIGLU>(p, stream) : CR_ERR_ARG;
        case EPI_PATCH: return (p.res && p.group > 0) ? launch_t<EPI_PATCH>(p, stream) : CR_ERR_ARG;
        case EPI_F32: return launch_t<EPI_F32>(p, stream);
    }
    return CR_ERR_ARG;
}
