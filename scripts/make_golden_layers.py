#!/usr/bin/env python3
"""Stage-wise golden vectors of the 32-layer language model (build container only; ~35 GB of RAM, ~12 min on 8 cores).

The reference's own InternLM2ForCausalLM (InternVL/modeling_internlm2.py, eager, bf16, seed-0 synthetic weights) on the
300-token prompt of scripts/make_golden_full_depth.py, with a forward hook on every InternLM2DecoderLayer (:621-681): the
residual stream BEFORE layer 0 and AFTER each of the 32 layers at three prompt rows (first, middle, last), i.e. what
`output_hidden_states=True` collects (:916-918, 965-967) minus the final norm.  The same prompt then goes through the same
modules in fp32 (same bf16 weight values): per layer, the reference's OWN bf16-vs-fp32 distance is the yardstick a second bf16
implementation is measured against (tests/test_gpu_full_depth.py::test_llm_layerwise_error_budget).
Output: tests/golden/full_depth_layers.npz (data only: bf16 bit patterns + the per-layer yardstick).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs, REF, OUT  # noqa: E402
from make_golden_full_depth import build_llm, PROMPT_TOKENS  # noqa: E402

ROWS = [0, PROMPT_TOKENS // 2, PROMPT_TOKENS - 1]


def run(llm, emb):
    kept = []
    hooks = []
    first = llm.model.layers[0].register_forward_pre_hook(lambda m, args, kwargs: kept.append((args[0] if args else kwargs['hidden_states'])[0, ROWS].detach().float().clone()), with_kwargs=True)
    for layer in llm.model.layers:
        hooks.append(layer.register_forward_hook(lambda m, a, out: kept.append(out[0][0, ROWS].detach().float().clone())))
    with torch.no_grad():
        logits = llm(inputs_embeds=emb, use_cache=False, return_dict=True).logits[0, -1].float()
    first.remove()
    for h in hooks:
        h.remove()
    assert len(kept) == len(llm.model.layers) + 1
    return torch.stack(kept), logits                  # [33][3][4096]


def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    cfg = json.load(open(os.path.join(REF, 'InternVL', 'config.json')))
    t0 = time.time()
    g = torch.Generator().manual_seed(300)
    emb = (torch.randn(1, PROMPT_TOKENS, 4096, generator=g) * 0.02).to(torch.bfloat16)
    llm = build_llm(cfg, torch.bfloat16)
    h16, lg16 = run(llm, emb)
    print(f'[{time.time() - t0:.0f}s] bf16 pass done', flush=True)
    del llm
    llm32 = build_llm(cfg, torch.float32)
    h32, lg32 = run(llm32, emb.float())
    print(f'[{time.time() - t0:.0f}s] fp32 pass done', flush=True)
    bits = (h16.contiguous().view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16)
    assert int((h16.contiguous().view(torch.int32) & 0xFFFF).abs().max()) == 0
    rel = [float((h16[l].double() - h32[l].double()).norm() / h32[l].double().norm()) for l in range(h16.shape[0])]
    gold = {'layers.bf16_bits': bits, 'layers.rows': np.array(ROWS, dtype=np.int64), 'layers.ref_bf16_vs_fp32_rel_l2': np.array(rel),
            'layers.ref_rms': np.array([float(h16[l].pow(2).mean().sqrt()) for l in range(h16.shape[0])]),
            'logits.ref_bf16_vs_fp32_rel_l2': np.float64(float((lg16.double() - lg32.double()).norm() / lg32.double().norm())),
            'meta': np.frombuffer(json.dumps({'seed': 0, 'prompt_seed': 300, 'prompt_tokens': PROMPT_TOKENS, 'rows': ROWS, 'torch': torch.__version__}).encode(), dtype=np.uint8)}
    for l in (0, 1, 2, 4, 8, 16, 24, 32):
        print(f'  after layer {l:2d}: residual rms {gold["layers.ref_rms"][l]:.3f}, reference bf16 vs fp32 rel-L2 {rel[l]:.3e}')
    path = os.path.join(OUT, 'full_depth_layers.npz')
    np.savez_compressed(path, **gold)
    print('wrote', path, os.path.getsize(path), 'bytes', f'{time.time() - t0:.0f}s')


if __name__ == '__main__':
    main()
