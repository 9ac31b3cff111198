#!/usr/bin/env python3
"""Randomised screen of round 6's two rewritten kernels (development aid; the fixed cases live in tests/):
  * decode with RoPE + split folded into the attention kernel against the separate launch: two contexts on the same weights (fold on / off), random batches of 9..64 rows,
    random prompt lengths 1..900 (both sides of the 256-key split boundaries), several steps each: logits of every step and the cache rows the fold writes must be the same bits;
    the same with the e4m3-weight decode (its partial sums take the same path);
  * perceiver attention: the current kernel against round 1's (CR_PERCEIVER_ATTN_V1=1) on random tile counts (1..300: both sides of the 252-tile chunk), bit-equal.
usage: python scripts/fuzz_round6.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd.engine import Engine
from callireader_amd import synthetic

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = torch.Generator().manual_seed(606)
dims = ModelDims.reduced(vit_layers=1, llm_layers=2, rs_depth=2, vocab=8201)
sd = synthetic.make_state_dict(dims, parts=('llm', 'resampler'), seed=0)
os.environ['CR_DECODE_FOLD_ROPE'] = '1'
a = Engine(dims, max_pos=2048)
a.load_state_dict(sd); a.load_rope(); a.finalize()
os.environ['CR_DECODE_FOLD_ROPE'] = '0'
os.environ['CR_PERCEIVER_ATTN_V1'] = '1'
b = Engine(dims, max_pos=2048)
b.share_weights_from(a)
del os.environ['CR_PERCEIVER_ATTN_V1']
n_checked = 0
for fp8 in (False, True):
    if fp8:
        a.enable_fp8_decode(True)
        b.share_weights_from(a)                                   # the borrower shares again after the owner's switch
    for r in range(R):
        rows = int(torch.randint(9, 65, (1,), generator=rng))
        mode = r % 3
        lens = [int(x) for x in torch.randint(1, (900, 300, 40)[mode], (rows,), generator=rng)]
        if mode == 0:
            lens[0], lens[1 % rows], lens[2 % rows] = 255, 256, 511                       # the new token lands on / next to a split boundary
        embs = [(torch.randn(S, 4096, generator=rng) * 0.02).to(torch.bfloat16).cuda() for S in lens]
        kvs = []
        for e in (a, b):
            kv = e.kv_alloc(rows, 1024)
            for i0 in range(0, rows, 16):
                idx = list(range(i0, min(rows, i0 + 16)))
                e.prefill_batch(kv, idx, [embs[i] for i in idx], penalty=1.3)
            kvs.append(kv)
        order = [int(x) for x in torch.randperm(rows, generator=rng)]
        for step in range(4):
            la = a.decode(kvs[0], order, penalty=1.3, want_logits=True)
            lb = b.decode(kvs[1], order, penalty=1.3, want_logits=True)
            torch.cuda.synchronize()
            assert torch.equal(la, lb), (fp8, r, rows, step, float((la - lb).abs().max()))
            n_checked += rows
        for i in (0, rows // 2, rows - 1):                                                # the cache rows the folded kernel wrote (last position of three sequences, both layers)
            pos = kvs[0].length(i) - 1
            for layer in (0, 1):
                for which in (0, 1):
                    assert torch.equal(kvs[0].read(layer, i, pos, which), kvs[1].read(layer, i, pos, which)), (fp8, r, i, layer, which)
        assert [kvs[0].generated(i) for i in range(rows)] == [kvs[1].generated(i) for i in range(rows)]
        for kv in kvs:
            kv.free()
print(f'decode fold: {n_checked} row-steps over {2 * R} random batches (bf16 and e4m3 weights): logits, cache rows and ids bit-equal')
gd = torch.Generator(device='cuda').manual_seed(7)
tot = 0
for r in range(R):
    T = int(torch.randint(1, 301, (1,), generator=rng))
    if r == 0:
        T = 253
    feats = (torch.randn(T, 256, 4096, device='cuda', generator=gd) * 0.7).bfloat16()
    x, y = a.resample(feats), b.resample(feats)
    torch.cuda.synchronize()
    assert torch.equal(x, y), (r, T)
    tot += T
print(f'perceiver attention: {tot} tiles over {R} random batches: bit-equal to round 1\'s kernel')
print('FUZZ_ROUND6 OK')
