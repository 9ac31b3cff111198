// Development aid: what does the ADDRESS PATTERN of one LDS-DMA instruction (global_load_lds, 16 bytes per lane) cost on gfx950?
//   A  16 rows x 64 B   (gemm256.hip today: lane -> row lane >> 2, chunk lane & 3)
//   B   8 rows x 128 B  (whole cache lines of a row-major operand)
//   C   1 KiB contiguous (a pre-tiled operand)
// Rows are ROWB bytes apart (2 KiB = K 1024, 8 KiB = K 4096).  Every wave of a 512-thread workgroup issues NI instructions back to back into its own
// LDS area from an L2-resident region, stamps s_memtime before / after the issue and after s_waitcnt vmcnt(0).  One workgroup per CU (160 KiB of LDS asked for).
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_pattern ldsdma_pattern.hip ; run: ./ldsdma_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define GLB(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int NI = 16;

template <int PAT>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int rowb, int reps, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this workgroup's region: 256 rows of rowb bytes; wave w works on rows 32 w .. 32 w + 31
    const char* base0 = src + ((size_t)blockIdx.x * 256) * rowb;       // (the eight waves read the same 32 rows: 64 KiB per workgroup, 2 MiB per XCD stay in its L2)
    unsigned long long issue = 0, land = 0;
    for (int r = 0; r < reps; r++) {
        unsigned long long t0, t1, t2;
        const char* base = base0 + (r % 4) * 512;       // walk the whole row over the repetitions: every cache set is used
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const char* a;
            if (PAT == 0) a = base + (size_t)((i & 1) * 16 + (lane >> 2)) * rowb + (i >> 1) * 64 + (lane & 3) * 16;        // 16 rows x 64 B, walking k
            else if (PAT == 1) a = base + (size_t)((i & 3) * 8 + (lane >> 3)) * rowb + (i >> 2) * 128 + (lane & 7) * 16;  // 8 rows x 128 B
            else a = base0 + (size_t)(r % 4) * 16384 + (size_t)i * 1024 + lane * 16;                                                                  // contiguous KiB
            __builtin_amdgcn_global_load_lds(GLB(a), LDS(smem + wave * (NI * 1024) + i * 1024), 16, 0, 0);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        if (r > 0) { issue += t1 - t0; land += t2 - t0; }
    }
    if (lane == 0) {
        out[((size_t)blockIdx.x * 8 + wave) * 2] = issue / (reps - 1);
        out[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = land / (reps - 1);
    }
}

int main() {
    const int grid = 256, reps = 129;
    char* src;
    const size_t bytes = (size_t)grid * 256 * 8192;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    unsigned long long* out;
    CK(hipMalloc(&out, grid * 8 * 2 * 8));
    std::vector<unsigned long long> h(grid * 8 * 2);
    const char* names[3] = {"A 16 rows x 64 B", "B 8 rows x 128 B", "C contiguous KiB"};
    for (int rowb : {2048, 8192})
        for (int pat = 0; pat < 3; pat++) {
            auto kern = pat == 0 ? k<0> : pat == 1 ? k<1> : k<2>;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, (const char*)src, rowb, reps, out);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> is, la;
            for (size_t i = 0; i < h.size(); i += 2) { is.push_back((double)h[i] / NI); la.push_back((double)h[i + 1]); }
            std::sort(is.begin(), is.end()); std::sort(la.begin(), la.end());
            printf("row stride %5d B, %-18s: issue %6.1f clocks per instruction (median wave; 8 waves per CU issuing together), all %d landed after %7.0f clocks\n", rowb,
                   names[pat], is[is.size() / 2], NI, la[la.size() / 2]);
        }
    return 0;
}
