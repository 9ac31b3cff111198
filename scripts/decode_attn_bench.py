#!/usr/bin/env python3
"""The batched decode attention alone (development aid): cr_op_decode_attention at the bench's context length, the streaming kernel (attention_decode.hip) against the matrix-core
split kernel (attention.hip), caches rotated past the Infinity Cache.   python scripts/decode_attn_bench.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
ROWS = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 64]
CTX, MAXT = 3200, 3328
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
for M in ROWS:
    ncopies = max(2, int(700e6 // (M * 8 * CTX * 128 * 4)) + 1)
    caches = [((torch.randn(M, 8, MAXT, 128, device=dev, generator=g)).bfloat16(), (torch.randn(M, 8, MAXT, 128, device=dev, generator=g)).bfloat16()) for _ in range(ncopies)]
    q = (torch.randn(M, 4096, device=dev, generator=g) * 0.7).bfloat16()
    seqs = torch.arange(M, device=dev, dtype=torch.int32)
    lens = torch.full((M,), CTX - 1, device=dev, dtype=torch.int32)
    res = {}
    from callireader_amd import _binding as B
    scratch = torch.empty(int(B.lib.cr_op_decode_attention_scratch_floats(M, CTX)), device=dev, dtype=torch.float32)
    out = torch.empty(M, 4096, device=dev, dtype=torch.bfloat16)
    for which, name in ((0, 'streaming'), (1, 'matrix-core split')):
        for kc, vc in caches: E.op_decode_attention(q, kc, vc, seqs, lens, which=which, max_keys=CTX, scratch=scratch, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(4):
                for kc, vc in caches: E.op_decode_attention(q, kc, vc, seqs, lens, which=which, max_keys=CTX, scratch=scratch, out=out)
            ev[1].record(); torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]) / (4 * ncopies) * 1e3)
        res[name] = best
    mb = M * 8 * CTX * 128 * 2 * 2 / 1e6
    print(f'rows {M:3d} ({mb:7.1f} MB of K / V, split + combine launches): ' + ', '.join(f'{k} {v:7.2f} us ({mb / v:.2f} TB/s)' for k, v in res.items()), flush=True)
