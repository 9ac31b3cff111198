#!/usr/bin/env python3
"""Same-process A/B of the decode step with RoPE + split folded into the attention kernel (ctx flag fold_rope, CR_DECODE_FOLD_ROPE at cr_create) against the separate
launch: two contexts over the SAME weights (cr_share_weights) and the same KV cache, alternating bursts of steps, medians over rounds.  ROWS=9,13,16,32,64"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from callireader_amd.config import ModelDims
from callireader_amd.engine import Engine
from callireader_amd.modeling_internvl_chat import InternVLChatModel

ROWS = [int(x) for x in os.environ.get('ROWS', '9,13,16,32,64').split(',')]
STEPS, ROUNDS = int(os.environ.get('STEPS', '48')), int(os.environ.get('ROUNDS', '7'))
dev = torch.device('cuda', 0)
dims = ModelDims.full()
S = bench.PAGE_TILES * 256 + bench.CHAR_TILES * 3 + bench.TEXT_TOKENS
P = max(ROWS)
os.environ['CR_DECODE_FOLD_ROPE'] = '1'
model = InternVLChatModel.from_synthetic(dims, seed=0, device=0, max_tokens=S + 4096, max_pages=P, parts=('llm',))
a = model.engine
os.environ['CR_DECODE_FOLD_ROPE'] = '0'
b = Engine(dims, device=0, max_pos=a.max_pos)
b.share_weights_from(a)
g = torch.Generator(device='cuda').manual_seed(1)
kv = a.kv_alloc(P, S + 4096)
kv.reset()
for i0 in range(0, P, 16):
    idx = list(range(i0, min(P, i0 + 16)))
    a.prefill_batch(kv, idx, [(torch.randn(S, dims.llm_hidden, device=dev, generator=g) * 0.02).bfloat16() for _ in idx], penalty=1.0)
out = {}
for n in ROWS:
    live = list(range(n))
    ts = {'fold': [], 'separate': []}
    for e in (a, b):
        for _ in range(4): e.decode(kv, live)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for name, e in (('fold', a), ('separate', b)) if r % 2 == 0 else (('separate', b), ('fold', a)):
            t0 = time.perf_counter()
            for _ in range(STEPS): e.decode(kv, live)
            torch.cuda.synchronize()
            ts[name].append((time.perf_counter() - t0) / STEPS * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    out[n] = {'fold_ms': round(med['fold'], 4), 'separate_ms': round(med['separate'], 4), 'gain_pct': round(100 * (1 - med['fold'] / med['separate']), 2),
              'cached_tokens_row0': kv.length(0)}
    print(f'rows {n:3d}: folded {med["fold"]:.3f} ms, separate launch {med["separate"]:.3f} ms ({out[n]["gain_pct"]:+.2f} %)   [{", ".join(f"{x:.3f}" for x in ts["fold"])}] vs [{", ".join(f"{x:.3f}" for x in ts["separate"])}]', flush=True)
print('RESULT ' + json.dumps(out))
