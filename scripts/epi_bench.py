#!/usr/bin/env python3
"""Epilogue cost microbenchmark (development aid): the ViT's three K=1024 / K=4096 GEMM shapes with each epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
def run(name, fn, flops, n=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    print(f'{name:32s} {ms:8.3f} ms {flops / ms / 1e9:8.1f} TFLOP/s', flush=True)
M = 64575
for (N, K) in [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
    bias = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    scale = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    res = (torch.rand(M, N, device='cuda', generator=g)).bfloat16()
    fl = 2.0 * M * N * K
    for kern in [int(x) for x in os.environ.get('KERNS', '2').split(',')]:
        run(f'k{kern} N={N} K={K} store nobias', lambda: E.op_gemm(0, A, W, kernel=kern), fl)
        run(f'k{kern} N={N} K={K} store bias', lambda: E.op_gemm(0, A, W, bias=bias, kernel=kern), fl)
        run(f'k{kern} N={N} K={K} gelu bias', lambda: E.op_gemm(1, A, W, bias=bias, kernel=kern), fl)
        run(f'k{kern} N={N} K={K} ls_res bias', lambda: E.op_gemm(2, A, W, bias=bias, scale=scale, res=res, kernel=kern), fl)
