// Development aid: issue cost of vector instructions on gfx950 at 1..4 waves per SIMD.
// Each wave runs REP x 64 independent instructions of one kind (8 register chains) between two s_memtime stamps;
// printed: cycles per instruction seen by a wave, and per SIMD (= that / waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP 256

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY64(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)

template <int OP>
__global__ void k(unsigned long long* out, float seed) {
    extern __shared__ char smem[];
    float r[8], q[8];
    for (int i = 0; i < 8; i++) { r[i] = seed + i + threadIdx.x; q[i] = seed * 0.5f + i; }
    float c = seed * 1.0001f;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 p[8];
    for (int i = 0; i < 8; i++) p[i] = f2{r[i], q[i]};
    f2 pc = {c, c};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < REP; it++) {
        if (OP == 0) {
#define I(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[n]) : "v"(c), "v"(q[n]));
            BODY64(I)
#undef I
        } else if (OP == 1) {
#define I(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[n]) : "v"(pc), "v"(pc));
            BODY64(I)
#undef I
        } else if (OP == 2) {
#define I(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[n]) : "v"(pc));
            BODY64(I)
#undef I
        } else if (OP == 3) {
#define I(n) asm volatile("v_exp_f32 %0, %0" : "+v"(r[n]));
            BODY64(I)
#undef I
        } else if (OP == 4) {
#define I(n) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[n]) : "v"(q[n]));
            BODY64(I)
#undef I
        } else if (OP == 5) {
#define I(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[n]) : "v"(q[n]), "v"(c));
            BODY64(I)
#undef I
        } else if (OP == 6) {
#define I(n) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r[n]));
            BODY64(I)
#undef I
        } else if (OP == 7) {
#define I(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[n]) : "v"(c));
            BODY64(I)
#undef I
        } else if (OP == 8) {
#define I(n) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[n]) : "v"(q[n]), "v"(c));
            BODY64(I)
#undef I
        } else if (OP == 9) {
#define I(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[n]) : "v"(pc));
            BODY64(I)
#undef I
        } else if (OP == 10) {      // mixed stream of the softmax: cvt, fma, exp, add  (x16)
#define I(n) asm volatile("v_cvt_pk_bf16_f32 %0, 0, %0\n\tv_fma_f32 %0, %0, %2, %3\n\tv_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %0" : "+v"(r[n]), "+v"(q[n]) : "v"(c), "v"(c));
            BODY8(I) BODY8(I)
#undef I
        } else if (OP == 11) {
#define I(n) asm volatile("v_exp_f16 %0, %0" : "+v"(r[n]));
            BODY64(I)
#undef I
        } else if (OP == 12) {
#define I(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[n]) : "v"(c));
            BODY64(I)
#undef I
        } else if (OP == 13) {      // exp alternating with fma: does the transcendental unit overlap plain VALU?
#define I(n) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(r[n]), "+v"(q[n]) : "v"(c));
            BODY8(I) BODY8(I) BODY8(I) BODY8(I)
#undef I
        } else if (OP == 14) {      // exp followed by three plain ops
#define I(n) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %2\n\tv_add_f32 %1, %1, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(r[n]), "+v"(q[n]) : "v"(c));
            BODY8(I) BODY8(I)
#undef I
        } else if (OP == 15) {
#define I(n) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[n]));
            BODY64(I)
#undef I
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i] + q[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_iter) {
    unsigned long long* d;
    hipMalloc(&d, (1 + 256 * 16) * 8);
    printf("%-28s", name);
    for (int W = 1; W <= 4; W++) {
        hipMemset(d, 0, (1 + 256 * 16) * 8);
        hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * W), 100 * 1024, 0, d, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1 + 256 * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (int b = 0; b < 256; b++) for (int w = 0; w < 4 * W; w++) { sum += h[1 + b * 16 + w]; n++; }
        double per = sum / n / ((double)REP * instr_per_iter);
        printf("  W=%d: %6.2f/wave %6.2f/SIMD", W, per, per / W);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    printf("cycles (s_memtime ticks) per instruction: as seen by one wave, and per SIMD, at W waves per SIMD\n");
    run<0>("v_fma_f32", 64);
    run<7>("v_add_f32", 64);
    run<12>("v_mul_f32", 64);
    run<1>("v_pk_fma_f32", 64);
    run<2>("v_pk_add_f32", 64);
    run<9>("v_pk_mul_f32", 64);
    run<3>("v_exp_f32", 64);
    run<11>("v_exp_f16", 64);
    run<4>("v_cvt_pk_bf16_f32", 64);
    run<5>("v_max3_f32", 64);
    run<6>("v_lshlrev_b32", 64);
    run<15>("v_and_b32 (literal)", 64);
    run<8>("v_dot2c_f32_bf16", 64);
    run<10>("cvt+fma+exp+add", 64);
    run<13>("exp+fma", 64);
    run<14>("exp+fma+add+mul", 64);
    return 0;
}
