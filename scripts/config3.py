#!/usr/bin/env python3
"""BASELINE config 3 alone (development aid): one example-shaped page on one GPU, batch of one -- per-phase milliseconds.
   python scripts/config3.py [new_tokens]         (under rocprofv3 --kernel-trace --stats for the per-kernel view)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from callireader_amd.config import ModelDims, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID
from callireader_amd import synthetic
from callireader_amd.modeling_internvl_chat import InternVLChatModel

new_tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device('cuda', 0)
dims = ModelDims.full()
S_page = bench.PAGE_TILES * 256 + bench.CHAR_TILES * 3 + bench.TEXT_TOKENS
model = InternVLChatModel.from_synthetic(dims, seed=0, device=0, max_tokens=S_page + new_tokens + 64, max_pages=1)
model.img_context_token_id = IMG_CONTEXT_TOKEN_ID
eng = model.engine
page_px = synthetic.make_pixels(bench.PAGE_TILES, seed=10, device=dev)
char_px = synthetic.make_pixels(bench.CHAR_TILES, seed=20, device=dev)
ids = bench.build_ids(bench.PAGE_TILES, bench.CHAR_TILES, bench.TEXT_TOKENS, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, 1000).to(dev)

def run(stamps=None):
    def st():
        if stamps is not None:
            torch.cuda.synchronize(); stamps.append(time.perf_counter())
    st()
    v = model.extract_feature(page_px); st()
    r, _ = model.align_tiles(char_px); st()
    e = eng.embed_splice(ids, v, r.reshape(-1, 3, dims.llm_hidden), img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID)
    out = model.generate_pages([e], max_new_tokens=1, eos_token_id=None); st()          # prefill + first pick
    out = model.generate_pages([e], max_new_tokens=new_tokens, eos_token_id=None); st()  # prefill again + new_tokens - 1 decode steps
    return out

run(); torch.cuda.synchronize()
best = None
for _ in range(3):
    s = []
    run(s)
    d = [1e3 * (b - a) for a, b in zip(s, s[1:])]
    if best is None or sum(d) < sum(best): best = d
vit, align, prefill, gen = best
dec = (gen - prefill) / (new_tokens - 1)
print(f'config 3: page ViT {vit:.1f} ms | char tiles ViT+resampler+VQ {align:.1f} ms | splice+prefill {prefill:.1f} ms | decode {dec:.3f} ms/token '
      f'({14.72e3 / dec:.0f} GB/s of weights) | page {1e-3 * (vit + align + prefill + dec * (new_tokens - 1)):.4f} s')
