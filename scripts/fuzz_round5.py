#!/usr/bin/env python3
"""Randomised screen of round 5's two new paths (development aid; the fixed cases live in tests/):
  * cr_op_decode_attention: random batch sizes, cache slots and lengths (1 .. 6 000 keys): the streaming kernel against the matrix-core split kernel (one bf16 step) and a
    row's bits independent of its batch (every row launched alone must equal its bits in the batch);
  * the prefill's last-rows-only final layer against the all-rows path (CR_PREFILL_LAST_ROWS=0): random page counts and lengths, logits bit-equal.
usage: python scripts/fuzz_round5.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
from callireader_amd.config import ModelDims
from callireader_amd import synthetic

R = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = 'cuda'
rng = torch.Generator().manual_seed(1234)
gd = torch.Generator(device=dev).manual_seed(5)
n_slots, MAXT = 12, 6016
kc = torch.randn(n_slots, 8, MAXT, 128, device=dev, generator=gd).bfloat16()
vc = torch.randn(n_slots, 8, MAXT, 128, device=dev, generator=gd).bfloat16()
worst = 0.0
for r in range(R):
    Bn = int(torch.randint(1, n_slots + 1, (1,), generator=rng))
    slots = torch.randperm(n_slots, generator=rng)[:Bn]
    mode = r % 3
    hi = (6000, 700, 70)[mode]
    lens = torch.randint(0, hi, (Bn,), generator=rng)
    q = (torch.randn(Bn, 4096, device=dev, generator=gd) * 0.7).bfloat16()
    seqs = slots.to(torch.int32).to(dev)
    lens_d = torch.zeros(n_slots, dtype=torch.int32, device=dev)
    lens_d[slots.to(dev)] = lens.to(torch.int32).to(dev)
    a = E.op_decode_attention(q, kc, vc, seqs, lens_d)
    b = E.op_decode_attention(q, kc, vc, seqs, lens_d, which=1)
    torch.cuda.synchronize()
    assert torch.isfinite(a.float()).all()
    d = float((a.float() - b.float()).abs().max())
    worst = max(worst, d)
    assert d <= 2 ** -6, (r, d)
    i = int(torch.randint(0, Bn, (1,), generator=rng))
    alone = E.op_decode_attention(q[i:i + 1].contiguous(), kc, vc, seqs[i:i + 1].contiguous(), lens_d)
    torch.cuda.synchronize()
    assert torch.equal(alone[0], a[i]), (r, i)
print(f'decode attention: {R} random launches, streaming vs matrix-core max |d| {worst:.4g}, batch independence bit-exact')

from callireader_amd.engine import Engine
dims = ModelDims.reduced(vit_layers=1, llm_layers=2, rs_depth=1, vocab=4099)
sd = synthetic.make_state_dict(dims, parts=('llm',), seed=0)
eng = Engine(dims, max_pos=4096)
eng.load_state_dict(sd); eng.load_rope(); eng.finalize()
os.environ['CR_PREFILL_LAST_ROWS'] = '0'
ref = Engine(dims, max_pos=4096)
del os.environ['CR_PREFILL_LAST_ROWS']
ref.share_weights_from(eng)
for r in range(max(4, R // 6)):
    n = int(torch.randint(1, 9, (1,), generator=rng))
    lens = [int(x) for x in torch.randint(1, 700, (n,), generator=rng)]
    if r % 4 == 0:
        lens[0] = 2100                                    # a batch past 2048 rows: the 256x256 kernel
    embs = [(torch.randn(1, S, 4096, generator=rng) * 0.02).to(torch.bfloat16).to(dev) for S in lens]
    outs = []
    for e in (eng, ref):
        kv = e.kv_alloc(n, 2200)
        lg = e.prefill_batch(kv, list(range(n)), embs, want_logits=True).clone()
        st = e.decode(kv, list(range(n)), want_logits=True).clone()
        torch.cuda.synchronize()
        outs.append((lg, st))
        kv.free()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (r, lens)
print('prefill last-rows-only final layer: random batches bit-equal to the all-rows path')
