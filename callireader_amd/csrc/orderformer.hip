// OrderFormer: the reading-order scorer of the detected columns.
//   reference: models/model.py:206-233 (`Transformer`: nn.Linear(4, 256) -> nn.TransformerEncoder of 4 post-norm
//   nn.TransformerEncoderLayer(d_model 256, 8 heads, dim_feedforward 2048, ReLU, eps 1e-5) -> nn.Linear(256, 1)),
//   :419-484 (`predict`: up to 50 boxes, zero-padded, bf16; padding rows attend like any other row).
// Tiny (5 M parameters, <= 50 rows per page): the linear layers go through the shared GEMM launcher with the rows of
// all pages of a call stacked; attention (8 heads x 50 x 50, d = 32), residual + LayerNorm, ReLU and the scalar
// decoder are small kernels here.  Rounding points follow the eager bf16 module: every linear output, the attention
// output, the residual sums and the LayerNorm outputs are bf16; softmax and LayerNorm statistics are fp32.
#include <string>

#include "ctx.hpp"

namespace {

constexpr int OD = 256, OH = 8, OHD = 32, OFF = 2048, OKPAD = 64;

// boxes [rows][4] -> [rows][64] (the GEMM's K granule), zero padded
__global__ void of_pad_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * OKPAD) return;
    const int r = i / OKPAD, c = i % OKPAD;
    out[i] = c < 4 ? in[r * 4 + c] : f2bf(0.f);
}

// one workgroup per (page, head); thread = query row.  qkv rows: [q(256) | k(256) | v(256)], head h = columns 32h..32h+31
__global__ __launch_bounds__(64) void of_attn_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int L) {
    __shared__ float ks[64][OHD + 1], vs[64][OHD + 1];
    const int page = blockIdx.x, head = blockIdx.y, t = threadIdx.x;
    const bf16* base = qkv + (int64_t)page * L * 3 * OD + head * OHD;
    float q[OHD];
    if (t < L) {
#pragma unroll
        for (int d = 0; d < OHD; d++) {
            q[d] = bf2f(base[(int64_t)t * 3 * OD + d]);
            ks[t][d] = bf2f(base[(int64_t)t * 3 * OD + OD + d]);
            vs[t][d] = bf2f(base[(int64_t)t * 3 * OD + 2 * OD + d]);
        }
    }
    __syncthreads();
    if (t >= L) return;
    const float scale = 0.17677669529663687f;          // 1 / sqrt(32)
    float m = -INFINITY, l = 0.f, acc[OHD];
#pragma unroll
    for (int d = 0; d < OHD; d++) acc[d] = 0.f;
    for (int j = 0; j < L; j++) {                        // online softmax over the keys, fp32
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < OHD; d++) s += q[d] * ks[j][d];
        s *= scale;
        const float mn = fmaxf(m, s);
        const float a = __expf(m - mn), p = __expf(s - mn);
        l = l * a + p;
#pragma unroll
        for (int d = 0; d < OHD; d++) acc[d] = acc[d] * a + p * vs[j][d];
        m = mn;
    }
    bf16* o = out + ((int64_t)page * L + t) * OD + head * OHD;
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < OHD; d++) o[d] = f2bf(acc[d] * inv);
}

// y = LayerNorm(x) over 256 columns, one wave per row (x already holds residual + sublayer, bf16)
__global__ __launch_bounds__(256) void of_ln_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, const bf16* __restrict__ gamma,
                                                   const bf16* __restrict__ beta, int rows, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bf16x4 v = *(const bf16x4*)(in + (int64_t)row * OD + lane * 4);
    float x[4] = {bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
    const float mean = wave_sum(x[0] + x[1] + x[2] + x[3]) * (1.0f / OD);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) sq += (x[e] - mean) * (x[e] - mean);
    const float rstd = rsqrtf(wave_sum(sq) * (1.0f / OD) + eps);
    const bf16x4 g = *(const bf16x4*)(gamma + lane * 4), b = *(const bf16x4*)(beta + lane * 4);
    bf16x4 y;
#pragma unroll
    for (int e = 0; e < 4; e++) y[e] = f2bf((x[e] - mean) * rstd * bf2f(g[e]) + bf2f(b[e]));
    *(bf16x4*)(out + (int64_t)row * OD + lane * 4) = y;
}

__global__ void of_relu_kernel(bf16* __restrict__ x, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    bf16x8 v = *(bf16x8*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = bf2f(v[e]) > 0.f ? v[e] : f2bf(0.f);
    *(bf16x8*)(x + i * 8) = v;
}

// score = bf16(x . w + b) per row, returned as fp32
__global__ __launch_bounds__(256) void of_decode_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const bf16* __restrict__ b,
                                                       float* __restrict__ out, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bf16x4 v = *(const bf16x4*)(x + (int64_t)row * OD + lane * 4), ww = *(const bf16x4*)(w + lane * 4);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) s += bf2f(v[e]) * bf2f(ww[e]);
    s = wave_sum(s);
    if (lane == 0) out[row] = rbf(s + bf2f(b[0]));
}

int of_gemm(cr_ctx* c, int epi, const bf16* A, int64_t lda, const bf16* Wt, int64_t ldw, const bf16* bias, bf16* C, int64_t ldc,
            const bf16* res, int M, int N, int K, hipStream_t st) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = Wt; p.ldw = ldw; p.C = C; p.ldc = ldc; p.bias = bias; p.res = res; p.ldr = ldc; p.M = M; p.N = N; p.K = K;
    return ctx_gemm(c, epi, p, st);
}

}  // namespace

extern "C" int cr_orderformer(cr_ctx* c, const void* boxes, int B, int L, float* scores, void* stream) {
    if (!c || !boxes || !scores || B <= 0 || L <= 0 || L > 64) return cr_fail(CR_ERR_ARG, "cr_orderformer: bad argument (1 <= L <= 64)");
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const std::string P = "orderformer.";
    const DevTensor* ew = WT(c, P + "embedding.weight");
    if (!ew || ew->numel() != (int64_t)OD * 4) return cr_fail(CR_ERR_STATE, "cr_orderformer: load the orderformer.* weights first (embedding.weight [256,4])");
    // the embedding weight padded to the GEMM's K granule, derived once
    if (!WT(c, "derived.orderformer.emb")) {
        DevTensor t;
        t.dtype = CR_BF16; t.shape = {OD, OKPAD}; t.bytes = (size_t)OD * OKPAD * 2;
        CR_HIP(hipMalloc(&t.ptr, t.bytes));
        CR_HIP(hipMemsetAsync(t.ptr, 0, t.bytes, st));
        CR_HIP(hipMemcpy2DAsync(t.ptr, OKPAD * 2, ew->ptr, 4 * 2, 4 * 2, OD, hipMemcpyDeviceToDevice, st));
        c->w["derived.orderformer.emb"] = t;
    }
    const int rows = B * L;
    CR_TRY(ws_ensure(c, (size_t)rows * (OKPAD + 3 * OD + 3 * OD + OFF) * 2 + 8192));
    Arena ar(c->ws);
    bf16* xin = ar.take<bf16>((size_t)rows * OKPAD);
    bf16* x = ar.take<bf16>((size_t)rows * OD);
    bf16* t = ar.take<bf16>((size_t)rows * OD);
    bf16* att = ar.take<bf16>((size_t)rows * OD);
    bf16* qkv = ar.take<bf16>((size_t)rows * 3 * OD);
    bf16* ffh = ar.take<bf16>((size_t)rows * OFF);
    hipLaunchKernelGGL(of_pad_kernel, dim3((rows * OKPAD + 255) / 256), dim3(256), 0, st, (const bf16*)boxes, xin, rows);
    CR_TRY(of_gemm(c, EPI_STORE, xin, OKPAD, W(c, "derived.orderformer.emb"), OKPAD, W(c, P + "embedding.bias"), x, OD, nullptr, rows, OD, OKPAD, st));
    for (int l = 0;; l++) {
        const std::string lp = P + "transformer_encoder.layers." + std::to_string(l) + ".";
        const bf16* wi = W(c, lp + "self_attn.in_proj_weight");
        if (!wi) { if (l == 0) return cr_fail(CR_ERR_STATE, "cr_orderformer: no encoder layers loaded"); break; }
        const bf16 *bi = W(c, lp + "self_attn.in_proj_bias"), *wo = W(c, lp + "self_attn.out_proj.weight"), *bo = W(c, lp + "self_attn.out_proj.bias"),
                   *w1 = W(c, lp + "linear1.weight"), *b1 = W(c, lp + "linear1.bias"), *w2 = W(c, lp + "linear2.weight"), *b2 = W(c, lp + "linear2.bias"),
                   *g1 = W(c, lp + "norm1.weight"), *e1 = W(c, lp + "norm1.bias"), *g2 = W(c, lp + "norm2.weight"), *e2 = W(c, lp + "norm2.bias");
        if (!bi || !wo || !bo || !w1 || !b1 || !w2 || !b2 || !g1 || !e1 || !g2 || !e2) return cr_fail(CR_ERR_STATE, "cr_orderformer: layer %d incomplete", l);
        CR_TRY(of_gemm(c, EPI_STORE, x, OD, wi, OD, bi, qkv, 3 * OD, nullptr, rows, 3 * OD, OD, st));
        hipLaunchKernelGGL(of_attn_kernel, dim3(B, OH), dim3(64), 0, st, qkv, att, L);
        CR_TRY(of_gemm(c, EPI_RES, att, OD, wo, OD, bo, t, OD, x, rows, OD, OD, st));            // x + out_proj(attn)
        hipLaunchKernelGGL(of_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, t, x, g1, e1, rows, 1e-5f);
        CR_TRY(of_gemm(c, EPI_STORE, x, OD, w1, OD, b1, ffh, OFF, nullptr, rows, OFF, OD, st));
        hipLaunchKernelGGL(of_relu_kernel, dim3((unsigned)(((int64_t)rows * OFF / 8 + 255) / 256)), dim3(256), 0, st, ffh, (int64_t)rows * OFF / 8);
        CR_TRY(of_gemm(c, EPI_RES, ffh, OFF, w2, OFF, b2, t, OD, x, rows, OD, OFF, st));          // x + linear2(relu(linear1 x))
        hipLaunchKernelGGL(of_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, t, x, g2, e2, rows, 1e-5f);
    }
    const bf16 *wd = W(c, P + "decoder.weight"), *bd = W(c, P + "decoder.bias");
    if (!wd || !bd) return cr_fail(CR_ERR_STATE, "cr_orderformer: decoder weights missing");
    hipLaunchKernelGGL(of_decode_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, wd, bd, scores, rows);
    CR_HIP(hipGetLastError());
    return CR_OK;
}
