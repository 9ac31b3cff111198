#!/usr/bin/env python3
"""GEMM microbenchmark (development aid): times cr_op_gemm on the hot path's shapes with random data.
CR_GEMM_FORCE=128|256 selects the tile kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E

shapes = [(32800, 3072, 1024), (32800, 1024, 1024), (32800, 4096, 1024), (32800, 1024, 4096),
          (64575, 3072, 1024), (64575, 1024, 1024), (64575, 4096, 1024), (64575, 1024, 4096),
          (3164, 6144, 4096), (3164, 4096, 4096), (3164, 28672, 4096), (3164, 4096, 14336),
          (25312, 6144, 4096), (25312, 4096, 14336), (4096, 4096, 4096), (8192, 8192, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
g = torch.Generator(device='cuda').manual_seed(0)
for (M, N, K) in shapes:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = (torch.rand(N, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    E.op_gemm(0, A, W); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 10
    ev[0].record()
    for _ in range(n):
        E.op_gemm(0, A, W)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    print(f'{os.environ.get("CR_GEMM_FORCE", "auto"):>4} M={M:6d} N={N:6d} K={K:6d}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s', flush=True)
