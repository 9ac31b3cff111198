#!/usr/bin/env python3
"""Full-depth golden vectors from the REFERENCE's own modules (build container only; needs ~40 GB of RAM).

What runs (all on CPU, eager attention, the reference's classes imported from /root/reference):
  * InternVisionModel, 24 layers (InternVL/modeling_intern_vit.py:399-437) + InternVLChatModel.extract_feature
    (pixel_shuffle + mlp1, InternVL/modeling_internvl_chat.py:283-319) on 2 tiles, bf16;
  * InternLM2ForCausalLM, 32 layers, vocabulary 92 553 (InternVL/modeling_internlm2.py:1022-1110), bf16, on a 300-token
    `inputs_embeds` prompt, then 8 greedy steps by a hand loop over `forward` with the tuple cache (the loop the
    reference delegates to transformers 4.45.2: SURVEY.md 8c);
  * the same LLM prompt once more with the module in fp32 (same bf16 weight values): the reference's own
    bf16-vs-exact noise at full depth, which is what calibrates the logit tolerance of tests/test_gpu_full_depth.py.
Weights are `callireader_amd.synthetic` tensors (seeded, regenerated bit for bit on the GPU box); only DATA is stored:
sub-sampled outputs, logits as bf16 bit patterns, token ids, top-2 margins.

Usage:  python scripts/make_golden_full_depth.py        (~10 min on 8 cores)
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs, sample, flat, REF, OUT  # noqa: E402

SEED = 0
PROMPT_TOKENS = 300
STEPS = 8
# further prompts: (seed, tokens, steps); per step the top-16 logits and a strided sample are kept instead of the whole row
EXTRA_PROMPTS = [(301, 64, 12), (302, 513, 12)]


def bf16_bits(t):
    """bf16-representable fp32 tensor -> uint16 bit patterns (the reference's logits are a bf16 GEMM .float())."""
    f = t.detach().float().contiguous()
    b = f.view(torch.int32)
    assert int((b & 0xFFFF).abs().max()) == 0, 'logits are not bf16 values'
    return (b >> 16).to(torch.int16).numpy().view(np.uint16)


def build_llm(cfg, dtype):
    """The reference's InternLM2ForCausalLM allocated directly in `dtype` with parameter init skipped (7.7 B normal
    draws would be overwritten anyway), then filled tensor by tensor from the seeded generator."""
    from transformers.initialization import no_init_weights
    from InternVL.configuration_internlm2 import InternLM2Config
    from InternVL.modeling_internlm2 import InternLM2ForCausalLM
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    lcfg = copy.deepcopy(cfg['llm_config'])      # the config class edits the nested rope_scaling dict in place
    lcfg['attn_implementation'] = 'eager'
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with no_init_weights():
            llm = InternLM2ForCausalLM(InternLM2Config(**lcfg))
    finally:
        torch.set_default_dtype(old)
    llm = llm.eval()
    params = dict(llm.named_parameters())
    seen = set()
    with torch.no_grad():
        for k, v in synthetic.iter_state_dict(ModelDims.full(), parts=('llm',), seed=SEED):
            name = k[len('language_model.'):]
            params[name].copy_(v)              # bf16 values, exact in fp32 too
            seen.add(name)
    assert seen == set(params), set(params) ^ seen
    # the rotary cache is a non-persistent buffer built in __init__ under the default dtype of that moment: rebuild it
    # the way a normally constructed module (fp32 default) holds it
    for layer in llm.model.layers:
        layer.attention._init_rope()           # modeling_internlm2.py:310-336
    return llm


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    cfg = json.load(open(os.path.join(REF, 'InternVL', 'config.json')))
    dims = ModelDims.full()
    gold = {}
    t0 = time.time()

    # ---------------- vision: 24 layers + projector on 2 tiles ----------------
    from InternVL.configuration_intern_vit import InternVisionConfig
    from InternVL.modeling_intern_vit import InternVisionModel
    from InternVL.modeling_internvl_chat import InternVLChatModel
    import types
    vcfg = dict(cfg['vision_config'])
    vcfg['use_flash_attn'] = False
    assert vcfg['num_hidden_layers'] == dims.vit_layers == 24
    vit = InternVisionModel(InternVisionConfig(**vcfg)).to(torch.bfloat16).eval()
    sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1'), seed=SEED)
    vit.load_state_dict({k[len('vision_model.'):]: v for k, v in sd.items() if k.startswith('vision_model.')}, strict=True)
    mlp1 = nn.Sequential(nn.LayerNorm(4096), nn.Linear(4096, 4096), nn.GELU(), nn.Linear(4096, 4096)).to(torch.bfloat16)
    mlp1.load_state_dict({k[len('mlp1.'):]: v for k, v in sd.items() if k.startswith('mlp1.')})
    px = synthetic.make_pixels(2, seed=1)
    with torch.no_grad():
        last = vit(pixel_values=px, output_hidden_states=False, return_dict=True).last_hidden_state
        ns = types.SimpleNamespace(vision_model=vit, mlp1=mlp1, select_layer=-1, downsample_ratio=0.5, ps_version='v2')
        ns.pixel_shuffle = lambda x, scale_factor=0.5: InternVLChatModel.pixel_shuffle(ns, x, scale_factor)
        feat = InternVLChatModel.extract_feature(ns, px)
        # the same modules in fp32 (same bf16 weight values): the reference's own rounding noise at this depth
        vit32, mlp32 = vit.float(), mlp1.float()
        ns32 = types.SimpleNamespace(vision_model=vit32, mlp1=mlp32, select_layer=-1, downsample_ratio=0.5, ps_version='v2')
        ns32.pixel_shuffle = lambda x, scale_factor=0.5: InternVLChatModel.pixel_shuffle(ns32, x, scale_factor)
        feat32 = InternVLChatModel.extract_feature(ns32, px.float())
    gold.update(flat('vit24.last', sample(last)))
    gold.update(flat('vit24.feat', sample(feat, 16384)))
    gold['vit24.feat_rows'] = feat[:, :4, :].float().numpy()            # 2 x 4 whole rows
    rel = float((feat.double() - feat32.double()).norm() / feat32.double().norm())
    gold['vit24.ref_bf16_vs_fp32_rel_l2'] = np.float64(rel)
    print(f'[{time.time() - t0:.0f}s] vision done; reference bf16 vs fp32 rel-L2 {rel:.3e}', flush=True)
    del vit, mlp1, vit32, mlp32, sd

    # ---------------- language model: 32 layers, vocab 92 553 ----------------
    def make_prompt(seed, tokens):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(1, tokens, 4096, generator=g) * 0.02).to(torch.bfloat16)

    def greedy(llm, emb, steps, tag, keep_rows):
        """Hand loop over the reference's forward (what transformers 4.45.2 _sample does for these arguments)."""
        rows, toks, margins, second, top_i, top_v, strided = [], [], [], [], [], [], []
        with torch.no_grad():
            o = llm(inputs_embeds=emb, use_cache=True, return_dict=True)
            logits, past = o.logits, o.past_key_values
            if keep_rows:
                gold[f'{tag}.k0_last'] = past[0][0][0, :, -1, :].float().numpy()
                gold[f'{tag}.v31_first'] = past[31][1][0, :, 0, :].float().numpy()
            for s in range(steps + 1):
                row = logits[0, -1].float()
                if keep_rows:
                    rows.append(bf16_bits(row))
                top = torch.topk(row, 16)
                nxt = int(torch.argmax(row))
                assert nxt == int(top.indices[0]) or float(top.values[0]) == float(top.values[1])
                toks.append(nxt)
                margins.append(float(top.values[0] - top.values[1]))
                second.append(int(top.indices[1]) if int(top.indices[0]) == nxt else int(top.indices[0]))
                top_i.append(top.indices.numpy().astype(np.int64)); top_v.append(top.values.numpy())
                strided.append(bf16_bits(row[::8]))
                print(f'[{time.time() - t0:.0f}s] {tag} step {s}: id {nxt}, top-2 margin {margins[-1]:.4f}', flush=True)
                if s == steps:
                    break
                pos = torch.tensor([[past[0][0].shape[2]]])
                o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=past, position_ids=pos, use_cache=True, return_dict=True)
                logits, past = o.logits, o.past_key_values
        if keep_rows:
            gold[f'{tag}.logits_bf16_bits'] = np.stack(rows)            # [steps + 1][92553]: prefill row, then one per fed id
        gold[f'{tag}.greedy_tokens'] = np.array(toks, dtype=np.int64)   # toks[s] = argmax of row s
        gold[f'{tag}.top2_margin'] = np.array(margins, dtype=np.float64)
        gold[f'{tag}.runner_up'] = np.array(second, dtype=np.int64)
        gold[f'{tag}.top16_ids'] = np.stack(top_i)
        gold[f'{tag}.top16_logits'] = np.stack(top_v)
        gold[f'{tag}.logits_stride8_bf16_bits'] = np.stack(strided)
        return rows

    emb = make_prompt(300, PROMPT_TOKENS)
    llm = build_llm(cfg, torch.bfloat16)
    print(f'[{time.time() - t0:.0f}s] bf16 LLM built', flush=True)
    rows = greedy(llm, emb, STEPS, 'llm32', True)
    for i, (seed, tokens, steps) in enumerate(EXTRA_PROMPTS):
        greedy(llm, make_prompt(seed, tokens), steps, f'llm32.extra{i}', False)
    row0 = torch.from_numpy(rows[0].astype(np.int32) << 16).view(torch.float32)
    del llm
    llm32 = build_llm(cfg, torch.float32)
    print(f'[{time.time() - t0:.0f}s] fp32 LLM built', flush=True)
    with torch.no_grad():
        l32 = llm32(inputs_embeds=emb.float(), use_cache=False, return_dict=True).logits[0, -1].float()
    d = (row0.double() - l32.double())
    gold['llm32.ref_bf16_vs_fp32_rel_l2'] = np.float64(float(d.norm() / l32.double().norm()))
    gold['llm32.ref_bf16_vs_fp32_max_abs'] = np.float64(float(d.abs().max()))
    gold['llm32.fp32_last_logits_sample'] = l32[::16].numpy()
    print(f'[{time.time() - t0:.0f}s] fp32 prefill done: reference bf16 vs fp32 rel-L2 '
          f'{float(gold["llm32.ref_bf16_vs_fp32_rel_l2"]):.3e}, max |d| {float(gold["llm32.ref_bf16_vs_fp32_max_abs"]):.3e}', flush=True)
    meta = {'seed': SEED, 'prompt_tokens': PROMPT_TOKENS, 'prompt_seed': 300, 'steps': STEPS, 'pixels_seed': 1, 'tiles': 2,
            'extra_prompts': EXTRA_PROMPTS,
            'torch': torch.__version__, 'threads': torch.get_num_threads()}
    gold['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, 'full_depth.npz')
    np.savez_compressed(path, **gold)
    print('wrote', len(gold), 'arrays,', os.path.getsize(path), 'bytes,', f'{time.time() - t0:.0f}s')


if __name__ == '__main__':
    main()
