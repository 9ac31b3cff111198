#!/usr/bin/env python3
"""Batched decode alone (development aid): P pages of the bench's prompt length prefilled once, then N decode steps timed.
   python scripts/decode_bench.py [pages] [steps]     (under rocprofv3 --kernel-trace --stats for the per-kernel view)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from callireader_amd.config import ModelDims, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID
from callireader_amd.modeling_internvl_chat import InternVLChatModel

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device('cuda', 0)
dims = ModelDims.full()
S = bench.PAGE_TILES * 256 + bench.CHAR_TILES * 3 + bench.TEXT_TOKENS
model = InternVLChatModel.from_synthetic(dims, seed=0, device=0, max_tokens=S + 256, max_pages=P)
eng = model.engine
g = torch.Generator(device='cuda').manual_seed(1)
embeds = [(torch.randn(S, dims.llm_hidden, device=dev, generator=g) * 0.02).bfloat16() for _ in range(P)]
kv = eng.kv_alloc(P, S + 256)
kv.reset()
for i0 in range(0, P, 16):
    idx = list(range(i0, min(P, i0 + 16)))
    eng.prefill_batch(kv, idx, [embeds[i] for i in idx], penalty=1.0)
live = list(range(P))
if os.environ.get('FP8'): eng.enable_fp8_decode(True)      # e4m3 copies of the LLM's linears for the decode GEMMs (an option)
for _ in range(4):
    eng.decode(kv, live, penalty=1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    eng.decode(kv, live, penalty=1.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f'decode, {P} pages at {S} + tokens: {dt * 1e3:.3f} ms per step')
