#!/usr/bin/env python3
"""Decode GEMM microbenchmark (development aid): the four InternLM2-7B projections at decode batch sizes, weights
alternated between two copies so that nothing is served from L2 / Infinity Cache."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
shapes = [('wqkv', 0, 6144, 4096), ('wo', 3, 4096, 4096), ('w1w3', 4, 28672, 4096), ('w2', 3, 4096, 14336)]
if os.environ.get('SLICED'):      # the K-sliced partial-sum kernels decode actually uses for wqkv / wo / w2
    shapes = [(n, 7 if n != 'w1w3' else e, N, K) for n, e, N, K in shapes]
for M in (int(a) for a in (sys.argv[1:] or ['32', '64'])):
    tot = 0.0
    for name, epi, N, K in shapes:
        A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
        Ws = [((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16() for _ in range(3)]
        res = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16) if epi == 3 else None
        out = E.op_gemm(epi, A, Ws[0], res=res)                    # preallocated: a fresh zero-filled output per call would be timed too
        which = {'wqkv': 0, 'wo': 1, 'w1w3': 2, 'w2': 3}[name]
        if os.environ.get('LAYOUT'):                               # the decode-layout copies cr_finalize keeps (wqkv's RoPE tile order needs the K-sliced form)
            kind = 2 if (which == 0 and epi == 7) else 1
            SW = {id(W): E.op_decode_swizzle(which if kind == 2 or which else 1, W) for W in Ws}
            f = lambda W: E.op_gemm(epi, A, W, res=res, out=out, decode_layout=(kind, SW[id(W)]))
        else:
            f = lambda W: E.op_gemm(epi, A, W, res=res, out=out)
        for W in Ws: f(W)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 30
        ev[0].record()
        for _ in range(n):
            for W in Ws: f(W)
        ev[1].record(); torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / (3 * n) * 1000
        tot += us
        print(f'M={M:3d} {name:5s} N={N:6d} K={K:6d}: {us:7.1f} us  {N * K * 2 / us / 1e6:5.2f} TB/s', flush=True)
    print(f'M={M:3d} layer total {tot:.1f} us -> {tot * 32 * 127 / 1000:.0f} ms per 127-token decode', flush=True)
