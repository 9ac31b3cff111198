// Development aid: what would ONE persistent launch per decoder layer buy over separate launches, for the weight streams alone?
// Four byte streams per layer with the sizes of InternLM2-7B's wqkv / wo / w1|w3 / w2 (50.3 / 33.6 / 234.9 / 117.4 MB, non-temporal 16-byte
// loads, two batches of eight per lane in flight, 512-thread workgroups), eight layer sets rotated (3.5 GB: nothing stays in the Infinity Cache):
//   A  one launch per stream (what gemm_decode.hip does today, minus the arithmetic)
//   B  one launch for all layers, a grid barrier after every stream (monotonic counter, or XCD-hierarchical), nothing in flight across it
//   C  the same, the first batch of the NEXT stream requested before the barrier (weight addresses do not depend on activations)
// Every spin is bounded: a barrier that does not complete sets an error flag and lets the kernel run out.
// build: hipcc --offload-arch=gfx950 -O3 -o persist_stream persist_stream.hip ; run: ./persist_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stream { const v4u* w; unsigned long long nb; };        // nb = batches of 512 threads x 8 x 16 B = 64 KiB
constexpr int UB = 8, NT = 512;

__device__ __forceinline__ void load_batch(v4u (&r)[UB], const v4u* w, unsigned long long b, int tid) {
    const v4u* p = w + b * (NT * UB) + tid;
#pragma unroll
    for (int u = 0; u < UB; u++) r[u] = __builtin_nontemporal_load(p + u * NT);
}
__device__ __forceinline__ void consume(const v4u (&r)[UB], unsigned& acc) {
#pragma unroll
    for (int u = 0; u < UB; u++) acc ^= r[u].x ^ r[u].w;
}

// this workgroup's batches of one stream: blockIdx.x, + gridDim.x, ...; `cur` may already hold the first one
__device__ __forceinline__ void run_stream(const Stream s, v4u (&cur)[UB], v4u (&nxt)[UB], bool have_first, int tid, unsigned& acc) {
    unsigned long long i = blockIdx.x;
    const unsigned long long G = gridDim.x;
    if (i >= s.nb) return;
    if (!have_first) load_batch(cur, s.w, i, tid);
    while (true) {
        const unsigned long long j = i + G;
        if (j < s.nb) load_batch(nxt, s.w, j, tid);
        consume(cur, acc);
        if (j >= s.nb) break;
        const unsigned long long k = j + G;
        if (k < s.nb) load_batch(cur, s.w, k, tid);
        consume(nxt, acc);
        if (k >= s.nb) break;
        i = k;
    }
}

__global__ __launch_bounds__(NT, 2) void one_stream(const Stream s, unsigned* sink) {
    v4u cur[UB], nxt[UB];
    unsigned acc = 0;
    run_stream(s, cur, nxt, false, threadIdx.x, acc);
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// The same bytes through gemm_decode.hip's addressing: the stream is an [N][K] bf16 matrix, a workgroup owns 16-row tiles (blockIdx.x, + gridDim.x, ...),
// wave kp walks k range kp * K / 8 in 32-steps, lane -> (row lane & 15, 16-byte k group lane >> 4): one load instruction = 16 rows x 64 contiguous bytes.
struct Rows { const unsigned short* w; int N, K; };
__device__ __forceinline__ void load_rows(v4u (&r)[UB], const Rows m, long idx, int nbt, int lane, int kp) {
    const long tile = blockIdx.x + (idx / nbt) * (long)gridDim.x;
    const int b = (int)(idx % nbt);
    const unsigned short* p = m.w + (tile * 16 + (lane & 15)) * (long)m.K + kp * (m.K / 8) + b * (UB * 32) + (lane >> 4) * 8;
#pragma unroll
    for (int u = 0; u < UB; u++) r[u] = __builtin_nontemporal_load((const v4u*)(p + u * 32));
}
__global__ __launch_bounds__(NT, 2) void rows_stream(const Rows m, unsigned* sink) {
    v4u cur[UB], nxt[UB];
    unsigned acc = 0;
    const int lane = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const int nbt = m.K / 8 / 32 / UB;                           // batches per tile and wave
    const long ntiles = m.N / 16;
    const long mine = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long total = mine * nbt;
    if (total == 0) return;
    long i = 0;
    load_rows(cur, m, 0, nbt, lane, kp);
    while (true) {
        if (i + 1 < total) load_rows(nxt, m, i + 1, nbt, lane, kp);
        consume(cur, acc);
        if (i + 1 >= total) break;
        if (i + 2 < total) load_rows(cur, m, i + 2, nbt, lane, kp);
        consume(nxt, acc);
        if (i + 2 >= total) break;
        i += 2;
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

struct Bar { unsigned* top; unsigned* xc; unsigned* xg; int* err; };

template <int KIND>   // 0: one monotonic counter; 1: per-XCD counters + a top counter + per-XCD generation words
__device__ __forceinline__ void grid_barrier(const Bar b, unsigned gen /* 1, 2, ... */) {
    __syncthreads();
    if (threadIdx.x == 0) {
        long spins = 0;
        if (KIND == 0) {
            __threadfence();
            atomicAdd(b.top, 1u);
            const unsigned target = gen * gridDim.x;
            while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { *b.err = 1; break; }
            }
            __threadfence();
        } else {
            const unsigned x = blockIdx.x & 7, per = (gridDim.x + 7 - x) / 8;       // workgroups with this blockIdx.x % 8
            __threadfence();
            const unsigned old = atomicAdd(b.xc + x * 32, 1u);
            if (old == gen * per - 1) {                          // last of this XCD: up to the top, wait for all eight, release the XCD
                atomicAdd(b.top, 1u);
                while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000) { *b.err = 1; break; }
                }
                __hip_atomic_store(b.xg + x * 32, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(b.xg + x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000) { *b.err = 1; break; }
                }
            }
            __threadfence();
        }
    }
    __syncthreads();
}

template <int KIND, bool PREFETCH>
__global__ __launch_bounds__(NT, 2) void persistent(const Stream* streams, int nstreams, const Bar bar, unsigned* sink) {
    v4u cur[UB], nxt[UB];
    unsigned acc = 0;
    const int tid = threadIdx.x;
    bool have = false;
    for (int s = 0; s < nstreams; s++) {
        const Stream st = streams[s];
        run_stream(st, cur, nxt, have, tid, acc);
        have = false;
        if (PREFETCH && s + 1 < nstreams) {
            const Stream nx = streams[s + 1];
            if (blockIdx.x < nx.nb) { load_batch(cur, nx.w, blockIdx.x, tid); have = true; }
        }
        grid_barrier<KIND>(bar, (unsigned)(s + 1));
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main() {
    const size_t mb[4] = {50331648, 33554432, 234881024, 117440512};
    const int SETS = 8, LAYERS = 32;
    std::vector<Stream> hs;
    std::vector<void*> bufs;
    for (int set = 0; set < SETS; set++)
        for (int k = 0; k < 4; k++) {
            void* p;
            CK(hipMalloc(&p, mb[k]));
            CK(hipMemset(p, set + k, mb[k]));
            bufs.push_back(p);
        }
    for (int l = 0; l < LAYERS; l++)
        for (int k = 0; k < 4; k++) hs.push_back({(const v4u*)bufs[(l % SETS) * 4 + k], mb[k] / (NT * UB * 16)});
    Stream* ds;
    CK(hipMalloc(&ds, hs.size() * sizeof(Stream)));
    CK(hipMemcpy(ds, hs.data(), hs.size() * sizeof(Stream), hipMemcpyHostToDevice));
    unsigned *sink, *ctr;
    int* err;
    CK(hipMalloc(&sink, 4096 * 4));
    CK(hipMalloc(&ctr, 4096 * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double layer_mb = (mb[0] + mb[1] + mb[2] + mb[3]) / 1e6;
    auto report = [&](const char* what, int grid, float ms) {
        const double us_layer = ms * 1e3 / LAYERS;
        printf("%-58s grid %4d: %7.1f us per layer  (%.2f TB/s)\n", what, grid, us_layer, layer_mb / us_layer);
    };
    {   // per stream, contiguous against GEMM addressing (grid = one tile per workgroup for N = 4096 / 6144, persistent 512 for w1|w3)
        const int NN[4] = {6144, 4096, 28672, 4096}, KK[4] = {4096, 4096, 4096, 14336};
        const char* nm[4] = {"wqkv", "wo", "w1|w3", "w2"};
        for (int k = 0; k < 4; k++) {
            for (int grid : {256, 384, 512}) {
                float ta = 1e9f, tb = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipEventRecord(e0));
                    for (int l = 0; l < LAYERS; l++) hipLaunchKernelGGL(one_stream, dim3(grid), dim3(NT), 0, 0, hs[l * 4 + k], sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < ta) ta = ms;
                    CK(hipEventRecord(e0));
                    for (int l = 0; l < LAYERS; l++) hipLaunchKernelGGL(rows_stream, dim3(grid), dim3(NT), 0, 0, Rows{(const unsigned short*)hs[l * 4 + k].w, NN[k], KK[k]}, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < tb) tb = ms;
                }
                printf("%-6s %6.1f MB grid %3d: contiguous %6.2f us (%.2f TB/s) | 16 rows x 64 B per instruction %6.2f us (%.2f TB/s)\n", nm[k], mb[k] / 1e6, grid,
                       ta * 1e3 / LAYERS, mb[k] / 1e6 / (ta * 1e3 / LAYERS), tb * 1e3 / LAYERS, mb[k] / 1e6 / (tb * 1e3 / LAYERS));
            }
        }
    }
    for (int grid : {256, 512}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            for (const Stream& s : hs) hipLaunchKernelGGL(one_stream, dim3(grid), dim3(NT), 0, 0, s, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        report("A  one launch per stream", grid, best);
        auto run = [&](const char* what, auto kern) {
            float b2 = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipMemsetAsync(ctr, 0, 4096 * 4));
                Bar bar{ctr, ctr + 64, ctr + 64 + 8 * 32, err};
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, 0, (const Stream*)ds, (int)hs.size(), bar, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < b2) b2 = ms;
            }
            int herr = 0;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            report(what, grid, b2);
            if (herr) { printf("   ^ a barrier timed out (grid not co-resident?)\n"); CK(hipMemset(err, 0, 4)); }
        };
        run("B  persistent, counter barrier, nothing across it", persistent<0, false>);
        run("C  persistent, counter barrier, next stream's first batch", persistent<0, true>);
        run("B' persistent, XCD barrier, nothing across it", persistent<1, false>);
        run("C' persistent, XCD barrier, next stream's first batch", persistent<1, true>);
    }
    return 0;
}
