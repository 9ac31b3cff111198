#!/usr/bin/env python3
"""Development aid: the e4m3 x e4m3 instance of the 256x256 kernel against the bf16 one on the shapes cr_enable_fp8_mfma moves
(ViT QKV / fc1 at M = 64 575, LLM prefill wqkv / w1|w3 at M = 50 624), and the norm kernels with bf16 and e4m3 output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
def run(name, fn, flops=None, nbytes=None, n=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    extra = f'{flops / ms / 1e9:8.1f} TFLOP/s' if flops else f'{nbytes / ms / 1e6:8.1f} GB/s'
    print(f'{name:44s} {ms:8.3f} ms {extra}', flush=True)
for (M, N, K, epi, what) in [(64575, 3072, 1024, 0, 'ViT QKV'), (64575, 4096, 1024, 1, 'ViT fc1+GELU'), (50624, 6144, 4096, 0, 'LLM wqkv'),
                             (50624, 28672, 4096, 4, 'LLM w1|w3 SwiGLU'), (16128, 4096, 4096, 1, 'mlp1[1]+GELU')]:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
    bias = None if epi == 4 else (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    a8, a_s = E.op_quantize_fp8(A)
    w8, w_s = E.op_quantize_fp8(W)
    fl = 2.0 * M * N * K
    run(f'{what} M={M} N={N} K={K} bf16', lambda: E.op_gemm(epi, A, W, bias=bias), fl)
    run(f'{what} M={M} N={N} K={K} e4m3', lambda: E.op_gemm_fp8x8(epi, a8, a_s, w8, w_s, bias=bias), fl)
    del A, W, a8, w8
for (rows, n, kind) in [(64575, 1024, 'ln'), (50624, 4096, 'rms')]:
    x = (torch.rand(rows, n, device='cuda', generator=g) * 2 - 1).bfloat16()
    gamma = torch.ones(n, device='cuda', dtype=torch.bfloat16)
    beta = torch.zeros(n, device='cuda', dtype=torch.bfloat16) if kind == 'ln' else None
    if kind == 'ln':
        run(f'LayerNorm {rows}x{n} -> bf16', lambda: E.op_layernorm(x, gamma, beta, 1e-6), nbytes=rows * n * 4)
    else:
        run(f'RMSNorm {rows}x{n} -> bf16', lambda: E.op_rmsnorm(x, gamma, 1e-6), nbytes=rows * n * 4)
    run(f'{kind} {rows}x{n} -> e4m3 + scale', lambda: E.op_norm_fp8(x, gamma, beta, 1e-6), nbytes=rows * n * 3)
