#!/usr/bin/env python3
"""Full-depth parity check (one-off, GPU box): the test-suite compares HIP and oracle on 1-2 layer models to stay fast;
this runs the REAL depths once -- InternViT 24 layers + mlp1 on 2 tiles, InternLM2 32 layers with the 92 553-row
vocabulary on a 96-token prompt + 8 teacher-forced decode steps -- and prints how the difference between the HIP path
and the oracle (CPU, bf16 eager) compares with the oracle's own bf16-vs-fp32 noise at that depth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from callireader_amd.engine import Engine
from oracle import vision, generate, internlm2

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())

torch.set_num_threads(min(32, os.cpu_count() or 8))
dims = ModelDims.full()
t0 = time.time()
# ---- vision ----
sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1'), seed=0)
eng = Engine(dims, max_pos=2048)
eng.load_state_dict(sd)
px = synthetic.make_pixels(2, seed=1)
with torch.no_grad():
    ref = vision.extract_feature(sd, px, dims.vit_layers)
    ref32 = vision.extract_feature({k: v.float() for k, v in sd.items()}, px.float(), dims.vit_layers)
print(f'[{time.time() - t0:.0f}s] oracle vision done', flush=True)
# ---- language model ----
lsd = synthetic.make_state_dict(dims, parts=('llm',), seed=0)
eng.load_state_dict(lsd)
eng.load_rope()
eng.finalize()
got = eng.extract_feature(px.cuda()).float().cpu()
print(f'extract_feature (24 layers + mlp1, 2 tiles): HIP vs oracle rel-L2 {rel(got, ref.float()):.3e};  oracle bf16 vs fp32 {rel(ref.float(), ref32):.3e}', flush=True)
g = torch.Generator().manual_seed(3)
emb = (torch.randn(1, 96, 4096, generator=g) * 0.02).to(torch.bfloat16)
with torch.no_grad():
    ids, logits = generate.greedy_generate(lsd, dims.llm_layers, emb, max_new_tokens=8, eos_token_id=-1, return_logits=True)
ids = ids[0].tolist()
print(f'[{time.time() - t0:.0f}s] oracle LLM done, ids {ids}', flush=True)
kv = eng.kv_alloc(1, 256)
lg = eng.prefill(kv, 0, emb.cuda(), want_logits=True)
kv2 = kv
worst, diverged = 0.0, []
for t in range(len(ids)):
    gotl, refl = lg.float().cpu().reshape(-1), logits[t]
    worst = max(worst, rel(gotl, refl))
    picked = kv.generated(0)[t]
    if picked != ids[t]:
        diverged.append((t, picked, ids[t], float(refl[ids[t]] - refl[picked])))
    if t + 1 < len(ids):
        lg = eng.decode(kv, [0], force_tokens=torch.tensor([ids[t]]), want_logits=True)
with torch.no_grad():
    lsd32 = {k: v.float() for k, v in lsd.items()}
    _, logits32 = generate.greedy_generate(lsd32, dims.llm_layers, emb.float(), max_new_tokens=1, eos_token_id=-1, return_logits=True)
    del lsd32
kv.reset(0)
lg0 = eng.prefill(kv, 0, emb.cuda(), want_logits=True).float().cpu().reshape(-1)
print(f'prefill logits: HIP vs oracle-bf16 {rel(lg0, logits[0]):.3e}; oracle-bf16 vs oracle-fp32 {rel(logits[0], logits32[0]):.3e}; HIP vs oracle-fp32 {rel(lg0, logits32[0]):.3e}', flush=True)
print(f'InternLM2 32 layers, vocab {dims.vocab}: logits HIP vs oracle worst rel-L2 over 8 steps {worst:.3e}; max |d logit| last step {float((gotl - refl).abs().max()):.3e};'
      f' greedy picks differing from the oracle: {diverged if diverged else "none"}', flush=True)
