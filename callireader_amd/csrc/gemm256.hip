// 256x256x64 bf16 MFMA GEMM for gfx950, 8 waves, LDS-DMA kept in flight across barriers.
//   C = epi(A[M,K] . W[N,K]^T),  K % 128 == 0  (K-tiles are processed in pairs)
//
// Geometry: 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 8 x 4 accumulators of
// v_mfma_f32_16x16x32_bf16 (128 VGPRs).  Per K-tile (64 deep) a wave runs 4 phases of 16 MFMAs, one
// 64 x 32 quadrant (mh, nh) of its sub-tile each: (0,0) (0,1) (1,1) (1,0), so consecutive phases reuse
// either the A or the B fragments already in registers.
//
// LDS: 2 K-tile buffers x 4 units x 16 KiB = 128 KiB.  A unit holds what ONE phase's ds_reads consume for ALL
// waves: U0 = A rows of quadrant-row mh=0 (of both wave rows), U1 = B cols of nh=0 (of all four wave columns),
// U2 = B nh=1, U3 = A mh=1.  A unit is 16 sub-tiles of 1 KiB = one MFMA operand fragment (16 rows x 32 k,
// 64-B rows); one global_load_lds_dwordx4 wave-instruction fills one sub-tile, 8 waves x 2 instructions fill
// a unit.  Inside a sub-tile the 16-B chunk of rows 8..15 is XORed with 2 (on the DMA source address and on
// the ds_read_b128 address), which makes the fragment read bank-conflict free.
//
// Schedule (one iteration = K-tiles 2i [even buffer] and 2i+1 [odd buffer], 8 phases):
//   phase   reads (ds_read_b128 -> regs)        MFMA quadrant     DMA issued (2 per wave)
//     1     even U0 (8) + U1 (4)                 (0,0)             odd  U3  of K-tile 2i+1
//     2     even U2 (4)                          (0,1)             even U0  of K-tile 2i+2
//     3     even U3 (8)                          (1,1)             even U1
//     4     -                                    (1,0)             even U2      then s_waitcnt vmcnt(6)
//     5     odd  U0 + U1                         (0,0)             even U3  of K-tile 2i+2
//     6     odd  U2                              (0,1)             odd  U0  of K-tile 2i+3
//     7     odd  U3                              (1,1)             odd  U1
//     8     -                                    (1,0)             odd  U2      then s_waitcnt vmcnt(6)
//   * vmcnt(6) leaves the three most recent units in flight: at phase 4 everything issued up to phase 1 has
//     landed = the whole odd K-tile, read from phase 5 on; at phase 8 the whole even K-tile of the next iteration.
//     The wait sits before the phase's closing barrier, the first read one phase later (RAW via wait + barrier).
//   * a unit is re-staged one phase (U0) or more after its last ds_read; those reads were retired by the
//     lgkmcnt(0) ahead of that phase's MFMAs and every wave has passed the closing barrier (WAR).
//   * the main loop never drains vmcnt to 0; barriers are raw s_barrier (a __syncthreads() would drain the DMA).
//   * measured nulls (kept out of the code): a second barrier at the start of each phase (-2..3 %), dropping the
//     closing barriers of phases 2/6 (hazard-free, +-0), compiler-placed fine-grained lgkmcnt waits instead of
//     lgkmcnt(0) (+-0), write-through (sc1) output stores (-1 %), skipping the epilogue entirely (< 7 %).
//
// Persistent: <= 256 workgroups (one per CU) walk the tile list.  The look-ahead of the schedule (1.75 K-tiles)
// runs straight through a tile boundary: in the last K-tile pair of a tile, phases 2..8 already stage K-tiles 0
// and 1 of the workgroup's NEXT tile, so the next tile starts in exactly the prologue state and no CU ever sits
// in a cold-start load burst (measured: with every CU starting a tile at once, the 112 KB/CU prologue costs ~5 us
// per tile at ~11 B/clk/CU; PMC: MFMA-busy 46 % at K = 1024 vs 61 % at K = 8192 before this change).
// The epilogue therefore may not touch the K buffers: it stages 16-row slices through a separate 4 KiB per wave.
#include <stdlib.h>

#include "gemm_epilogue.hpp"

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int UNIT = 16384;                       // bytes per unit
constexpr int KBUF = 4 * UNIT;                    // bytes per K-tile buffer
constexpr int LDS_MAIN2 = 2 * KBUF;               // 131072
constexpr int LDS_BYTES2 = LDS_MAIN2 + 8 * 4096;  // + one 16x64 fp32 slice per wave = 163840 (all of the CU's LDS)

#define WAIT_VM6() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#define WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)   /* lgkmcnt(0); the builtin (unlike inline asm) is seen by the compiler's own wait insertion */

// rows [row0, row0+16) x cols [col0, col0+64) from the wave-private slice st[16][64]
template <int EPI>
__device__ __forceinline__ void epilogue_rows16(const GemmParams& p, const float* st, int row0, int col0, int lane) {
    if (EPI == EPI_SWIGLU) {
        const int row = lane >> 2, oc = lane & 3;
        const int gm = row0 + row, gno = col0 / 2 + oc * 8;
        if (gm < p.M && gno + 8 <= p.N / 2) {
            const float* sp = st + row * 64 + oc * 16;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float g = rbf(sp[e]), u = rbf(sp[8 + e]);
                o[e] = f2bf(rbf(silu(g)) * u);
            }
            *(bf16x8*)((bf16*)p.C + (int64_t)gm * p.ldc + gno) = o;
        }
        return;
    }
    const bool vec_ok = ((p.ldc & 7) == 0) || (EPI == EPI_F32);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int row = it * 8 + (lane >> 3);
        const int c8 = (lane & 7) * 8;
        const int gm = row0 + row, gn = col0 + c8;
        if (gm >= p.M || gn >= p.N) continue;
        const float* sp = st + row * 64 + c8;
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
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int ntm = (p.M + BM2 - 1) / BM2, ntn = (p.N + BN2 - 1) / BN2;
    const int ntiles = ntm * ntn;
    const int nk = p.K / BK2;

    // tile list index -> (m0, n0): bijective XCD remap (workgroup b and b + 8k share an XCD), then 8-row super-groups
    auto tile_origin = [&](int orig, int& m0, int& n0) {
        const int xcd = orig & 7, q = ntiles >> 3, r = ntiles & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        constexpr int GM = 8;
        const int per_group = GM * ntn;
        const int grp = pid / per_group;
        const int first_m = grp * GM;
        const int gsz = min(ntm - first_m, GM);
        const int in_g = pid - grp * per_group;
        m0 = (first_m + in_g % gsz) * BM2;
        n0 = (in_g / gsz) * BN2;
    };

    // DMA sources: this wave fills sub-tiles s = 2*wave + e (e = 0,1) of every unit
    //   A units: s -> (wave row s>>3, fragment (s>>1)&3, ksub s&1);  B units: s -> (wave col s>>2, fragment (s>>1)&1, ksub s&1)
    //   lane -> row lane>>2 of the sub-tile, 16-B chunk (lane&3) ^ 2*(row>>3)
    // kept as 32-bit byte offsets from the (uniform) operand base: operands are < 4 GiB, and the fragment double
    // buffer below needs the registers
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((srow >> 3) << 1);
    uint32_t qA[2];           // [mh]   offsets used by phases 2..8 (may already be the NEXT tile's); piece e = 1 is 64 B on
    uint32_t qB[2];           // [nh]
    uint32_t p1A;             // U3 (A mh=1) of the tile being computed, for phase 1
    auto make_ptrs = [&](int m0, int n0) {
        // sub-tiles 2*wave and 2*wave + 1 are the two k-halves (ksub 0 / 1) of the same 16 rows
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int arow = m0 + (wave >> 2) * 128 + h * 64 + (wave & 3) * 16 + srow;
            const int brow = n0 + (wave >> 1) * 64 + h * 32 + (wave & 1) * 16 + srow;
            qA[h] = (uint32_t)(((int64_t)min(arow, p.M - 1) * p.lda + schunk * 8) * 2);
            qB[h] = (uint32_t)(((int64_t)min(brow, p.N - 1) * p.ldw + schunk * 8) * 2);
        }
    };
    auto dma = [&](const bf16* base, uint32_t o, int kt, int buf, int u) {
        char* dst = smem + buf * KBUF + u * UNIT + (2 * wave) * 1024;
        const char* src = (const char*)base + (int64_t)kt * (BK2 * 2);
        __builtin_amdgcn_global_load_lds(CR_GLB(src + o), CR_LDS(dst), 16, 0, 0);
        // the instruction's immediate offset is added to the global AND to the LDS address: M0 is set 64 short
        __builtin_amdgcn_global_load_lds(CR_GLB(src + o), CR_LDS(dst + 1024 - 64), 16, 64, 0);
    };

    const int lrow = lane & 15;
    const int lane_off = lrow * 64 + (((lane >> 4) ^ ((lrow >> 3) << 1)) * 16);
    const int a_sub = wm * 8 * 1024 + lane_off;          // + (i*2 + ksub) * 1024
    const int b_sub = wn * 4 * 1024 + lane_off;          // + (j*2 + ksub) * 1024
    float* st = (float*)(smem + LDS_MAIN2) + wave * 1024;

    f32x4 acc[8][4];
    bf16x8 ra0[4][2], ra1[4][2];      // A fragments [i][ksub]: mh = 0 lives in ra0, mh = 1 in ra1
    bf16x8 rbx[2][2], rby[2][2];      // B fragments [j][ksub]: the two halves swap registers every K-tile

#define READ_A(dst, buf, unit)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++) \
        dst[i][ks] = *(const bf16x8*)(smem + (buf) * KBUF + (unit) * UNIT + a_sub + (i * 2 + ks) * 1024);
#define READ_B(dst, buf, unit)                                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++) \
        dst[j][ks] = *(const bf16x8*)(smem + (buf) * KBUF + (unit) * UNIT + b_sub + (j * 2 + ks) * 1024);
    // one phase: operands of THIS phase were read a phase ago (retired by the lgkmcnt(0)); the reads for the NEXT
    // phase and this phase's two DMA pieces are issued ahead of the 16 MFMAs and complete underneath them
#define PHASE(mh, nh, RA, RB, PREFETCH, DMA)                                                    \
    WAIT_LGKM0();                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    PREFETCH;                                                                                   \
    DMA;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    __builtin_amdgcn_s_setprio(1);                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ks++) _Pragma("unroll") for (int i = 0; i < 4; i++) \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                           \
            acc[(mh) * 4 + i][(nh) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(RA[i][ks], RB[j][ks], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    WAIT_VM8();                                                                                 \
    __builtin_amdgcn_s_barrier();

    int t_cur = blockIdx.x;
    if (t_cur >= ntiles) return;
    int m0, n0;
    tile_origin(t_cur, m0, n0);
    make_ptrs(m0, n0);
    // ---- cold start (first tile of this workgroup only): K-tile 0 complete, U0..U2 of K-tile 1 in flight ----
    dma(p.A, qA[0], 0, 0, 0); dma(p.W, qB[0], 0, 0, 1); dma(p.W, qB[1], 0, 0, 2); dma(p.A, qA[1], 0, 0, 3);
    dma(p.A, qA[0], 1, 1, 0); dma(p.W, qB[0], 1, 1, 1); dma(p.W, qB[1], 1, 1, 2);
    WAIT_VM6();
    __builtin_amdgcn_s_barrier();
    READ_A(ra0, 0, 0); READ_B(rbx, 0, 1);

    while (true) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t_next = t_cur + gridDim.x;
        const bool has_next = t_next < ntiles;

        for (int kt = 0; kt < nk; kt += 2) {
            p1A = qA[1];
            int k2 = kt + 2;                                   // K-tile staged by phases 2..5 (and k2 + 1 by 6..8)
            PHASE(0, 0, ra0, rbx, READ_B(rby, 0, 2), dma(p.A, p1A, kt + 1, 1, 3));
            if (k2 >= nk) {                                    // last pair: the look-ahead belongs to the next tile
                if (has_next) { int nm0, nn0; tile_origin(t_next, nm0, nn0); make_ptrs(nm0, nn0); k2 = 0; }
                else k2 = nk - 2;                              // nothing follows: re-load dead units with valid addresses
            }
            PHASE(0, 1, ra0, rby, READ_A(ra1, 0, 3), dma(p.A, qA[0], k2, 0, 0));
            PHASE(1, 1, ra1, rby, READ_A(ra0, 1, 0), dma(p.W, qB[0], k2, 0, 1));
            PHASE(1, 0, ra1, rbx, READ_B(rby, 1, 1), dma(p.W, qB[1], k2, 0, 2));
            PHASE(0, 0, ra0, rby, READ_B(rbx, 1, 2), dma(p.A, qA[1], k2, 0, 3));
            PHASE(0, 1, ra0, rbx, READ_A(ra1, 1, 3), dma(p.A, qA[0], k2 + 1, 1, 0));
            PHASE(1, 1, ra1, rbx, READ_A(ra0, 0, 0), dma(p.W, qB[0], k2 + 1, 1, 1));
            PHASE(1, 0, ra1, rby, READ_B(rbx, 0, 1), dma(p.W, qB[1], k2 + 1, 1, 2));
        }

        // ---- epilogue: eight 16-row slices per wave through its private 4 KiB (the K buffers stay untouched) ----
#pragma unroll
        for (int mf = 0; mf < 8; mf++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) st[((lane >> 4) * 4 + e) * 64 + j * 16 + (lane & 15)] = acc[mf][j][e];
            __builtin_amdgcn_wave_barrier();
            epilogue_rows16<EPI>(p, st, m0 + wm * 128 + mf * 16, n0 + wn * 64, lane);
            __builtin_amdgcn_wave_barrier();
        }
        if (!has_next) break;
        t_cur = t_next;
        tile_origin(t_cur, m0, n0);
    }
    WAIT_VM0();       // dead re-loads of the final pair must land before the workgroup releases its LDS
}

template <int EPI>
int launch_t(const GemmParams& p, hipStream_t stream) {
    const int ntiles = ((p.M + BM2 - 1) / BM2) * ((p.N + BN2 - 1) / BN2);
    static int n_cu = 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES2) != hipSuccess)
            return CR_ERR_HIP;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CR_ERR_HIP;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), LDS_BYTES2, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

}  // namespace

bool gemm256_supported(int epi, const GemmParams& p) {
    if (p.M < 2048 || p.N < 512 || (p.K % 128) != 0) return false;
    // wave-quantisation model (calibrated on the measured shapes): a round of 256x256 tiles on
    // 256 CUs costs 1, a round of 512 co-resident 128x128 tiles 0.55; pick the kernel with the cheaper schedule
    const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const double c256 = (double)((t256 + 255) / 256);
    const double c128 = 0.55 * (double)((t128 + 511) / 512);
    return c256 <= c128;
}

int launch_gemm256(int epi, const GemmParams& p, hipStream_t stream) {
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE>(p, stream);
        case EPI_GELU: return launch_t<EPI_GELU>(p, stream);
        case EPI_LS_RES: return launch_t<EPI_LS_RES>(p, stream);
        case EPI_RES: return launch_t<EPI_RES>(p, stream);
        case EPI_SWIGLU: return launch_t<EPI_SWIGLU>(p, stream);
        case EPI_PATCH: return launch_t<EPI_PATCH>(p, stream);
        case EPI_F32: return launch_t<EPI_F32>(p, stream);
    }
    return CR_ERR_ARG;
}
