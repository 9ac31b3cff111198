// 256x128x32 bf16 MFMA GEMM for gfx950 with TWO independent 4-wave workgroups per CU.
//   C = epi(A[M,K] . W[N,K]^T),  K % 32 == 0
//
// Why a second tiled kernel.  gemm256.hip keeps one 8-wave workgroup per CU: its accumulators fill the register file and
// its K ring fills the LDS, so NOTHING runs beside a tile's epilogue.  In-kernel stamps at the ViT's K = 1024 shapes
// (scripts/gemm_stamps.py): fc1 + GELU spends 19.4 k of a tile's 62 k ticks in the epilogue (vector pipe at its two-waves
// throughput, matrix pipe idle), QKV + bias 7 k of 46 k, proj + LayerScale + residual 10 k of 53 k.  Here a CU holds two
// workgroups of four waves (one wave per SIMD each, 80 KiB of LDS each) that walk their own tile lists: while one is in its
// epilogue the other owns the matrix pipe.  Per wave nothing changes -- the same 128 x 64 sub-tile as 8 x 4 accumulators of
// v_mfma_f32_16x16x32_bf16 with the weights as the A operand, the same fragment images, the same epilogue code
// (gemm256_epilogue.hpp) -- but a workgroup covers 256 x 128 of C and steps K by 32:
//   K-tile = A 256 x 32 (16 sub-tiles of 1 KiB = one MFMA operand fragment each: 16 rows x 32 k, 64-B rows, the 16-B chunk of
//            rows 8..15 XORed with 2 on the DMA source and on the ds_read_b128) + W 128 x 32 (8 sub-tiles) = 24 KiB;
//   ring   = 3 K-tiles (72 KiB) + one 2-KiB epilogue slice per wave = 80 KiB;
//   a wave fills sub-tiles 6w .. 6w+5 of every K-tile (6 global_load_lds_dwordx4 pieces), three per phase;
//   phase  = memory slot (this phase's ds_read_b128s, three DMA pieces of K-tile kt+2, s_barrier)
//          + matrix slot (s_waitcnt lgkmcnt(0), 16 MFMAs at s_setprio 1, s_barrier); two phases per K-tile:
//          A: W fragments (4) + A fragments of rows 0..63 (4) -> acc[0..3][*];   B: A fragments of rows 64..127 (4) -> acc[4..7][*];
//   * RAW: K-tile kt+2 is staged during K-tile kt; "s_waitcnt vmcnt(6)" at the end of memory slot B(kt+1) leaves only
//     K-tile kt+3's six pieces in flight, and the first read of K-tile kt+2 comes two barriers later;
//   * WAR: K-tile kt+2 lands in the slot of K-tile kt-1, whose reads every wave retired (lgkmcnt(0)) before the barrier that
//     closes matrix slot B(kt-1);
//   * persistent: <= 2 x CUs workgroups walk the tile list; the look-ahead runs through tile boundaries (K-tiles 0 and 1 of the
//     next tile are staged by the last two K-tiles of this one) and is drained (vmcnt(0) + barrier) before the epilogue, so
//     the first counted wait of the next tile (skipped at its K-tile 0) never sits behind this tile's output stores.
// The two workgroups of a CU are not synchronised with each other; the hardware interleaves them (one wave of each per SIMD).
#include <stdlib.h>

#include "gemm256_epilogue.hpp"

namespace {

constexpr int BMW = 256, BNW = 128, BKW = 32;
constexpr int KSLOT = 24 * 1024;                 // 16 A sub-tiles + 8 W sub-tiles
constexpr int LDS_RING = 3 * KSLOT;              // 73728
constexpr int LDS_2W = LDS_RING + 4 * 2048;      // 81920 = half of the CU's LDS

#define W2_WAIT_VM6() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#define W2_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define W2_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm2w_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int ntm = (p.M + BMW - 1) / BMW, ntn = (p.N + BNW - 1) / BNW;
    const int ntiles = ntm * ntn;
    const int nk = p.K / BKW;

    // tile list index -> (m0, n0): bijective XCD remap (workgroups b and b + 8k share an XCD), then 8-row super-groups
    auto tile_origin = [&](int orig, int& m0, int& n0) {
        const int xcd = orig & 7, q = ntiles >> 3, r = ntiles & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        constexpr int GM = 8;
        const int per_group = GM * ntn;
        const int grp = pid / per_group;
        const int first_m = grp * GM;
        const int gsz = min(ntm - first_m, GM);
        const int in_g = pid - grp * per_group;
        m0 = (first_m + in_g % gsz) * BMW;
        n0 = (in_g / gsz) * BNW;
    };

    // DMA sources: this wave fills sub-tiles 6*wave + e (e = 0..5) of every K-tile; sub-tile s < 16 is A rows 16s..16s+15,
    // s >= 16 is W rows 16(s-16)..; lane -> row lane>>2 of the sub-tile, 16-B chunk (lane&3) ^ 2*(row>>3).
    // Kept as 32-bit byte offsets from the (uniform) operand base of the tile the LOOK-AHEAD is in.
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ ((srow >> 3) << 1);
    uint32_t off[6];
    auto make_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int e = 0; e < 6; e++) {
            const int s = wave * 6 + e;                        // wave-uniform
            if (s < 16) off[e] = (uint32_t)(((int64_t)min(m0 + s * 16 + srow, p.M - 1) * p.lda + schunk * 8) * 2);
            else off[e] = (uint32_t)(((int64_t)min(n0 + (s - 16) * 16 + srow, p.N - 1) * p.ldw + schunk * 8) * 2);
        }
    };
    auto dma = [&](int e, int kt, int slot) {
        const int s = wave * 6 + e;
        const char* base = (const char*)(s < 16 ? (const void*)p.A : (const void*)p.W) + (int64_t)kt * (BKW * 2);
        __builtin_amdgcn_global_load_lds(CR_GLB(base + off[e]), CR_LDS(smem + slot * KSLOT + s * 1024), 16, 0, 0);
    };

    const int lrow = lane & 15;
    const int lane_off = lrow * 64 + (((lane >> 4) ^ ((lrow >> 3) << 1)) * 16);
    const int a_sub = wm * 8 * 1024 + lane_off;            // + mf * 1024
    const int b_sub = (16 + wn * 4) * 1024 + lane_off;     // + j * 1024
    char* stg = smem + LDS_RING + wave * 2048;

    f32x4 acc[8][4];
    bf16x8 ra[4], rb[4];

#define W2_READ_A(slot, half) _Pragma("unroll") for (int i = 0; i < 4; i++) ra[i] = *(const bf16x8*)(smem + (slot) * KSLOT + a_sub + ((half) * 4 + i) * 1024);
#define W2_READ_B(slot) _Pragma("unroll") for (int j = 0; j < 4; j++) rb[j] = *(const bf16x8*)(smem + (slot) * KSLOT + b_sub + j * 1024);
#define W2_MFMA(half)                                                                            \
    W2_WAIT_LGKM0();                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_setprio(1);                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int j = 0; j < 4; j++)  \
        acc[(half) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[j], ra[i], acc[(half) * 4 + i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_barrier();

    int t_cur = blockIdx.x;
    if (t_cur >= ntiles) return;
    int m0, n0;
    tile_origin(t_cur, m0, n0);
    make_offsets(m0, n0);
    // ---- cold start (first tile of this workgroup): K-tiles 0 and 1 landed
#pragma unroll
    for (int e = 0; e < 6; e++) dma(e, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int e = 0; e < 6; e++) dma(e, 1, 1);
    }
    W2_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    int slot = 0;                                          // slot of the K-tile being multiplied
    bool first = true;

    while (true) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t_next = t_cur + gridDim.x;
        const bool has_next = t_next < ntiles;

        for (int kt = 0; kt < nk; kt++) {
            // the K-tile staged during this one: kt + 2 of this tile, or the next tile's first ones
            int la = kt + 2;
            if (la >= nk) {
                if (la == nk && has_next) { int nm0, nn0; tile_origin(t_next, nm0, nn0); make_offsets(nm0, nn0); }
                la = has_next ? la - nk : nk - 1;           // nothing follows: re-load a dead slot with bytes it may keep
            }
            const int slot_la = slot >= 1 ? slot - 1 : 2;  // (slot + 2) % 3
            // ---- phase A: rows 0..63 of the wave's sub-tile
            W2_READ_B(slot);
            W2_READ_A(slot, 0);
            dma(0, la, slot_la); dma(1, la, slot_la); dma(2, la, slot_la);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            W2_MFMA(0);
            // ---- phase B: rows 64..127
            W2_READ_A(slot, 1);
            dma(3, la, slot_la); dma(4, la, slot_la); dma(5, la, slot_la);
            __builtin_amdgcn_sched_barrier(0);
            if (kt != 0 || first) W2_WAIT_VM6();            // K-tile kt+1 has landed (in every wave once the barrier is passed)
            __builtin_amdgcn_s_barrier();
            W2_MFMA(1);
            slot = slot == 2 ? 0 : slot + 1;
        }
        first = false;
        W2_WAIT_VM0();                                      // the look-ahead (K-tiles 0, 1 of the next tile) has landed ...
        __builtin_amdgcn_s_barrier();                       // ... in every wave
        epilogue_tile<EPI, 1>(p, acc, stg, m0 + wm * 128, n0 + wn * 64, lane);
        if (!has_next) break;
        t_cur = t_next;
        tile_origin(t_cur, m0, n0);
    }
    W2_WAIT_VM0();
}

template <int EPI>
int launch_t(const GemmParams& p, hipStream_t stream) {
    const int ntiles = ((p.M + BMW - 1) / BMW) * ((p.N + BNW - 1) / BNW);
    static std::atomic<uint64_t> attr_done{0};
    if (!cr_dyn_lds_once(attr_done, (const void*)gemm2w_kernel<EPI>, LDS_2W)) return CR_ERR_HIP;
    const int slots = 2 * cr_device_cus();
    hipLaunchKernelGGL(gemm2w_kernel<EPI>, dim3(ntiles < slots ? ntiles : slots), dim3(256), LDS_2W, stream, p);
    return hipGetLastError() == hipSuccess ? CR_OK : CR_ERR_HIP;
}

}  // namespace

int launch_gemm2w(int epi, const GemmParams& p, hipStream_t stream) {
    if (p.K % BKW != 0 || p.K < 2 * BKW) return CR_ERR_ARG;
    switch (epi) {
        case EPI_STORE: return launch_t<EPI_STORE>(p, stream);
        case EPI_GELU: return launch_t<EPI_GELU>(p, stream);
        case EPI_LS_RES: return launch_t<EPI_LS_RES>(p, stream);
        case EPI_RES: return launch_t<EPI_RES>(p, stream);
        case EPI_SWIGLU: return launch_t<EPI_SWIGLU>(p, stream);
        case EPI_PATCH: return launch_t<EPI_PATCH>(p, stream);
        case EPI_F32: return launch_t<EPI_F32>(p, stream);
        case EPI_ARGMAX: return launch_t<EPI_ARGMAX>(p, stream);
    }
    return CR_ERR_ARG;
}
