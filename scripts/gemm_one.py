#!/usr/bin/env python3
"""One tiled-GEMM shape, a few launches (development aid: the program under `rocprofv3 --pmc ...` in scripts/traffic_clock.py).  usage: gemm_one.py M N K [epi]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
M, N, K = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = torch.Generator(device='cuda').manual_seed(0)
A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
bias = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
out = E.op_gemm(epi, A, W, bias=bias, kernel=2)
for _ in range(6):
    E.op_gemm(epi, A, W, bias=bias, kernel=2, out=out)
torch.cuda.synchronize()
