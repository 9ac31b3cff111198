#!/usr/bin/env python3
"""Round-4 verdict, item 3b: would decoding BOTH in-flight batches as one 128-row batch pay?  The weights would be read once per 128 rows instead of
twice (68 GB instead of 2 x 41.5 GB per 128 pages), if a kernel existed that multiplies 128 rows at the weight streams' rate.  The candidates that exist:
the tiled kernels (128x128 and 256x256 tiles, weights through LDS) at M = 128, against two launches of the 64-row weight-streaming kernels decode
uses today (K-sliced partials for wqkv / wo / w2, the X-through-LDS stream kernel for w1|w3), all on the decode-layout / nn.Linear weights they really read.
    python scripts/decode128_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
shapes = [('wqkv', 0, 6144, 4096), ('wo', 1, 4096, 4096), ('w1w3', 2, 28672, 4096), ('w2', 3, 4096, 14336)]

def timeit(fs, n=20):
    for f in fs: f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        for f in fs: f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / (len(fs) * n) * 1000

tot = {'two 64-row launches': 0.0, '128x128 tiles': 0.0, '256x256 tiles': 0.0}
for name, which, N, K in shapes:
    Ws = [((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16() for _ in range(3)]
    A64 = (torch.rand(64, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    A128 = (torch.rand(128, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    epi_small = 4 if name == 'w1w3' else 7                       # what decode launches: SwiGLU stream kernel / K-sliced partials
    kind = 2 if which == 0 else 1
    SW = [E.op_decode_swizzle(which if kind == 2 or which else 1, W) for W in Ws]
    out_s = E.op_gemm(epi_small, A64, Ws[0])
    t64 = timeit([(lambda W=W, S=S: E.op_gemm(epi_small, A64, W, out=out_s, decode_layout=(kind, S))) for W, S in zip(Ws, SW)])
    epi_big = 4 if name == 'w1w3' else 0
    res = {}
    for label, kern in (('128x128 tiles', 1), ('256x256 tiles', 2)):
        out_b = E.op_gemm(epi_big, A128, Ws[0], kernel=kern)
        res[label] = timeit([(lambda W=W: E.op_gemm(epi_big, A128, W, kernel=kern, out=out_b)) for W in Ws])
    tot['two 64-row launches'] += 2 * t64
    for k, v in res.items(): tot[k] += v
    print(f'{name:5s} N={N:6d} K={K:6d}: one 64-row launch {t64:7.1f} us ({N * K * 2 / t64 / 1e6:4.2f} TB/s), two {2 * t64:7.1f} | M = 128: 128x128 tiles {res["128x128 tiles"]:7.1f} us, 256x256 tiles {res["256x256 tiles"]:7.1f} us', flush=True)
print('one decoder layer\'s four GEMMs for 128 rows: ' + ', '.join(f'{k} {v:.1f} us' for k, v in tot.items()))
