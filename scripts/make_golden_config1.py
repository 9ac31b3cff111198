#!/usr/bin/env python3
"""BASELINE config 1 at FULL depth through the reference's own modules (build container only; ~15 min, ~30 GB).

The reference's example page (tests/golden/example0.jpg = examples/0.jpg, 788x2000, with its 96 labelled character
boxes) goes the way `chat_ocr(use_p=True, hard_vq=False, drop_zero=False, repetition_penalty=1.0)` sends it
(modeling_internvl_chat.py:649-762): 11 page tiles + 96 character tiles -> InternVisionModel (24 layers) + extract_feature
-> PerceiverResampler (depth 4) -> vq_cos_sim against the 92 553-row table -> the calli_align tail (the reference's own
statements, compiled out of the method) -> prompt of 3 164 ids from the reference's tokenizer -> embedding + the two masked
overwrites (:1087-1105) -> InternLM2ForCausalLM (32 layers, eager) prefill + 16 greedy steps of a hand loop (the loop the
reference delegates to transformers 4.45.2).  Seeded synthetic weights (no checkpoint exists offline).
Stored in tests/golden/config1_full_depth.npz: the prompt ids, VQ indices, samples of the visual / pseudo-token
embeddings, per step the greedy id, top-16 logits and a stride-8 sample of the row.  Data only.
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs, sample, flat, REF, OUT  # noqa: E402
from make_golden_full_depth import build_llm, bf16_bits  # noqa: E402

STEPS = 16
QUESTION = '这幅书法作品内容是什么？'                  # inference.py:69


def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic, preprocess
    from make_golden_tail import reference_tail
    from make_golden_tokenizer import patched_model_dir
    cfg = json.load(open(os.path.join(REF, 'InternVL', 'config.json')))
    dims = ModelDims.full()
    gold = {}
    t0 = time.time()
    img = Image.open(os.path.join(OUT, 'example0.jpg')).convert('RGB')
    boxes = preprocess.boxes_from_labelme(json.load(open(os.path.join(OUT, 'example0_boxes.json'))))
    page_px = preprocess.load_image(img).to(torch.bfloat16)                                  # pinned to the reference's tiling
    arr = np.array(img)
    char_px = torch.cat([preprocess.load_image_2(Image.fromarray(arr[y1:y2, x1:x2])).to(torch.bfloat16) for x1, y1, x2, y2 in boxes])
    assert page_px.shape[0] == 11 and char_px.shape[0] == 96

    # ---- vision ----
    from InternVL.configuration_intern_vit import InternVisionConfig
    from InternVL.modeling_intern_vit import InternVisionModel
    from InternVL.modeling_internvl_chat import InternVLChatModel
    vcfg = dict(cfg['vision_config'])
    vcfg['use_flash_attn'] = False
    vit = InternVisionModel(InternVisionConfig(**vcfg)).to(torch.bfloat16).eval()
    sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1'), seed=0)
    vit.load_state_dict({k[len('vision_model.'):]: v for k, v in sd.items() if k.startswith('vision_model.')}, strict=True)
    mlp1 = nn.Sequential(nn.LayerNorm(4096), nn.Linear(4096, 4096), nn.GELU(), nn.Linear(4096, 4096)).to(torch.bfloat16)
    mlp1.load_state_dict({k[len('mlp1.'):]: v for k, v in sd.items() if k.startswith('mlp1.')})
    ns = types.SimpleNamespace(vision_model=vit, mlp1=mlp1, select_layer=-1, downsample_ratio=0.5, ps_version='v2')
    ns.pixel_shuffle = lambda x, scale_factor=0.5: InternVLChatModel.pixel_shuffle(ns, x, scale_factor)
    with torch.no_grad():
        feat_chars = torch.cat([InternVLChatModel.extract_feature(ns, char_px[i:i + 8]) for i in range(0, 96, 8)])
        print(f'[{time.time() - t0:.0f}s] character tiles encoded', flush=True)
        feat_page = InternVLChatModel.extract_feature(ns, page_px)
    print(f'[{time.time() - t0:.0f}s] page tiles encoded', flush=True)
    del vit, mlp1, sd
    gold.update(flat('feat_page', sample(feat_page, 8192)))
    gold.update(flat('feat_chars', sample(feat_chars, 8192)))

    # ---- CalliAlign: resampler, VQ, tail ----
    from models.perceiver_resampler import PerceiverResampler
    from models.similarity import vq_cos_sim
    rs = PerceiverResampler(dim=4096, depth=dims.rs_depth).to(torch.bfloat16).eval()
    rsd = synthetic.make_state_dict(dims, parts=('resampler',), seed=0)
    rs.load_state_dict({k[len('resampler.'):]: v for k, v in rsd.items()}, strict=True)
    vsd = synthetic.make_state_dict(dims, parts=('vq',), seed=0)
    table = nn.Embedding(dims.vocab, 4096).to(torch.bfloat16)
    table.weight.data.copy_(vsd['normed_emb.weight'])
    with torch.no_grad():
        out = rs(feat_chars)
        indices = vq_cos_sim(table, out, False)
        # second-best cosine per query: how close the reference's own arg-max is to a tie
        xn = torch.nn.functional.normalize(out, p=2, dim=2)
        en = torch.nn.functional.normalize(table.weight, p=2, dim=1)
        sim = torch.matmul(xn, en.t()).float()
        top2 = torch.topk(sim, 2, dim=2).values
        top8 = torch.topk(sim, 8, dim=2)                 # the reference's own candidates: what the VQ tie rule (oracle/calli_align.py: vq_tie_rule) looks a differing index up in
        del sim
    tail, _ = reference_tail()
    self_ns = types.SimpleNamespace(normed_emb=types.SimpleNamespace(weight=table.weight.data), mu=vsd['calli.mu'], sigma=vsd['calli.sigma'])
    back, indices2 = tail(self_ns, indices, out.clone(), False, False, False)
    print(f'[{time.time() - t0:.0f}s] CalliAlign done: {tuple(back.shape)} pseudo-token rows', flush=True)
    gold.update(flat('resampler', sample(out, 8192)))
    gold['vq.indices'] = indices.numpy().astype(np.int64)
    gold['vq.top2_cos'] = top2.numpy()
    gold['vq.top8_ids'] = top8.indices.numpy().astype(np.int64)
    gold['vq.top8_cos'] = top8.values.numpy()
    gold.update(flat('pseudo', sample(back, 8192)))
    del rs, rsd, table, xn, en

    # ---- prompt (the reference's template + tokenizer) ----
    from InternVL.conversation import get_conv_template
    d, _ = patched_model_dir()
    from InternVL.tokenization_internlm2 import InternLM2Tokenizer
    tok = InternLM2Tokenizer.from_pretrained(d)
    question = '<image>\n' + QUESTION + '[UNUSED_TOKEN_140]' * back.shape[0]                # :690-699
    t = get_conv_template('internlm2-chat')
    t.append_message(t.roles[0], question)
    t.append_message(t.roles[1], None)
    query = t.get_prompt().replace('<image>', '<img>' + '<IMG_CONTEXT>' * 256 * 11 + '</img>', 1)   # :707-724
    ids = tok(query, return_tensors='pt')['input_ids']
    gold['input_ids'] = ids[0].numpy().astype(np.int64)
    print(f'[{time.time() - t0:.0f}s] prompt of {ids.shape[1]} ids', flush=True)

    # ---- LLM: embed + two masked overwrites (:1087-1105), prefill, greedy hand loop ----
    llm = build_llm(cfg, torch.bfloat16)
    with torch.no_grad():
        emb = llm.get_input_embeddings()(ids)
        B, N, C = emb.shape
        emb = emb.reshape(B * N, C)
        flat_ids = ids.reshape(B * N)
        sel = flat_ids == 92546
        assert int(sel.sum()) == 11 * 256
        emb[sel] = feat_page.reshape(-1, C)
        sel = flat_ids == 92537
        assert int(sel.sum()) == back.shape[0]
        emb[sel] = back.reshape(-1, C).to(emb.dtype)
        emb = emb.reshape(B, N, C)
        o = llm(inputs_embeds=emb, use_cache=True, return_dict=True)
        logits, past = o.logits, o.past_key_values
        print(f'[{time.time() - t0:.0f}s] prefill done', flush=True)
        toks, margins, top_i, top_v, strided = [], [], [], [], []
        for s in range(STEPS):
            row = logits[0, -1].float()
            top = torch.topk(row, 16)
            nxt = int(torch.argmax(row))
            toks.append(nxt)
            margins.append(float(top.values[0] - top.values[1]))
            top_i.append(top.indices.numpy().astype(np.int64)); top_v.append(top.values.numpy())
            strided.append(bf16_bits(row[::8]))
            print(f'[{time.time() - t0:.0f}s] step {s}: id {nxt}, top-2 margin {margins[-1]:.4f}', flush=True)
            if s == STEPS - 1:
                break
            pos = torch.tensor([[past[0][0].shape[2]]])
            o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=past, position_ids=pos, use_cache=True, return_dict=True)
            logits, past = o.logits, o.past_key_values
    gold['greedy_tokens'] = np.array(toks, dtype=np.int64)
    gold['top2_margin'] = np.array(margins)
    gold['top16_ids'] = np.stack(top_i)
    gold['top16_logits'] = np.stack(top_v)
    gold['logits_stride8_bf16_bits'] = np.stack(strided)
    gold['meta'] = np.frombuffer(json.dumps({'steps': STEPS, 'question': QUESTION, 'seed': 0, 'torch': torch.__version__}).encode(), dtype=np.uint8)
    path = os.path.join(OUT, 'config1_full_depth.npz')
    np.savez_compressed(path, **gold)
    print('wrote', path, os.path.getsize(path), 'bytes', f'{time.time() - t0:.0f}s')


if __name__ == '__main__':
    main()
