#!/usr/bin/env python3
"""Free-running greedy streams of the REFERENCE on the margin-controlled ("peaked") checkpoint (build container only;
~20 GB of RAM, ~25 min on 8 cores).

What runs: the reference's own InternLM2ForCausalLM (InternVL/modeling_internlm2.py:1022-1110; 32 layers, vocabulary 92 553,
eager attention, bf16) on `callireader_amd.synthetic.iter_peaked_llm` weights, driven the way
`language_model.generate(inputs_embeds=..., use_cache=True, repetition_penalty=..., num_beams=1, do_sample=False,
max_new_tokens=..., eos_token_id=92542)` is driven from InternVL/modeling_internvl_chat.py:1111-1120: prompt = the config-1
ids (3 158 ids of the reference's template and tokenizer for examples/0.jpg, tests/golden/config1_full_depth.npz) through
`tok_embeddings` + the two masked overwrites (:1087-1105), then a hand loop over `forward` with the tuple cache
(prepare_inputs_for_generation, modeling_internlm2.py:1112-1149).  transformers 4.45.2's `_sample` cannot run under the
installed 5.x; its three moving parts are taken from the INSTALLED transformers instead of being restated:
`RepetitionPenaltyLogitsProcessor`, `EosTokenCriteria`, `MaxLengthCriteria` (unchanged since 4.45.2).

Three streams, all free-running (each token fed back is the reference's own pick):
  A    prompt = config-1 ids,                       repetition_penalty 1.0, stops on EOS as its 72nd token;
  B15  prompt = the same with the last id replaced, repetition_penalty 1.5, seven steps at which the un-penalised arg-max is
       an id generated earlier (the penalty decides the pick), stops on EOS as its 80th token;
  B10  the B prompt, repetition_penalty 1.0, max_new_tokens 40: loops from the first such step on, stops on the length.
Stored in tests/golden/peaked_streams.npz: ids, per step the top-2 margin of the PROCESSED scores, the top-16 raw logits,
and the walk the checkpoint was built to produce (so a CPU test can check golden == construction).  Data only.

  python scripts/make_golden_peaked.py --calibrate     short prompt, a few steps: prints logits / margins (5 min)
  python scripts/make_golden_peaked.py                 the golden file
  python scripts/make_golden_peaked.py --long          tests/golden/peaked_long.npz (round 6): the LONG streams -- synthetic.PEAKED_LONG at 2 layers, full width,
                                                       full vocabulary: LA 1 024 free-running tokens (the API's default max_new_tokens, no EOS on the way), LB
                                                       EOS as its 85th token (85 = 5 mod 16) with eight steps the penalty decides, LC enters walk A 1 003 tokens
                                                       in and stops on A's EOS as its 397th token; all three at repetition_penalty 1.5 (chat_ocr's default,
                                                       modeling_internvl_chat.py:652), text-only prompts of 333 / 77 / 200 ids.  ids + margins only (< 100 KB).
"""
import argparse
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

SEED = 0


def build_peaked_llm(cfg, start_a, dims=None, pcfg=None):
    """As make_golden_full_depth.build_llm, filled from synthetic.iter_peaked_llm (dims / pcfg: the reduced-depth LONG variant)."""
    import copy
    from transformers.initialization import no_init_weights
    from InternVL.configuration_internlm2 import InternLM2Config
    from InternVL.modeling_internlm2 import InternLM2ForCausalLM
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    dims = dims or ModelDims.full()
    pcfg = pcfg or synthetic.PEAKED
    lcfg = copy.deepcopy(cfg['llm_config'])
    lcfg['attn_implementation'] = 'eager'
    lcfg['num_hidden_layers'] = dims.llm_layers
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with no_init_weights():
            llm = InternLM2ForCausalLM(InternLM2Config(**lcfg))
    finally:
        torch.set_default_dtype(old)
    llm = llm.eval()
    params = dict(llm.named_parameters())
    seen = set()
    with torch.no_grad():
        for k, v in synthetic.iter_peaked_llm(dims, start_a, seed=SEED, cfg=pcfg):
            name = k[len('language_model.'):]
            params[name].copy_(v)
            seen.add(name)
    assert seen == set(params), set(params) ^ seen
    for layer in llm.model.layers:
        layer.attention._init_rope()           # modeling_internlm2.py:310-336 (see make_golden_full_depth.build_llm)
    return llm


def run_stream(llm, emb, penalty, max_new_tokens, eos, tag, t0, past0=None, keep_top=True, quiet=False):
    """The greedy loop of transformers 4.45.2 `_sample` for these arguments, with the installed release's own processor and
    stopping criteria.  Returns (record dict, (logits, past) of the prefill so that a second stream can start from it)."""
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
    from transformers.generation.stopping_criteria import EosTokenCriteria, MaxLengthCriteria
    proc = RepetitionPenaltyLogitsProcessor(penalty=penalty) if penalty != 1.0 else None
    stop_eos, stop_len = EosTokenCriteria(eos_token_id=eos), MaxLengthCriteria(max_length=max_new_tokens)
    ids = torch.zeros(1, 0, dtype=torch.long)          # only inputs_embeds is given: generate() starts from an empty input_ids
    toks, margins, top_i, top_v, raw_gap = [], [], [], [], []
    with torch.no_grad():
        if past0 is None:
            o = llm(inputs_embeds=emb, use_cache=True, return_dict=True)
            past0 = (o.logits[:, -1, :].float().clone(), o.past_key_values)
            print(f'[{time.time() - t0:.0f}s] {tag}: prefill of {emb.shape[1]} rows done', flush=True)
        row, past = past0
        while True:
            scores = proc(ids, row.clone()) if proc is not None else row
            top = torch.topk(scores[0], 2)
            nxt = int(torch.argmax(scores[0]))
            raw = torch.topk(row[0], 16)
            toks.append(nxt)
            margins.append(float(top.values[0] - top.values[1]))
            raw_gap.append(float(row[0, nxt] - raw.values[0]))            # < 0 where the penalty overruled the raw arg-max
            if keep_top:
                top_i.append(raw.indices.numpy().astype(np.int64)); top_v.append(raw.values.numpy())
            ids = torch.cat([ids, torch.tensor([[nxt]])], dim=1)
            if not quiet or len(toks) % 64 == 0 or raw_gap[-1] < 0:
                print(f'[{time.time() - t0:.0f}s] {tag} token {len(toks)}: id {nxt}, margin {margins[-1]:.3f}, raw top {float(raw.values[0]):.2f}'
                      f'{" (penalty decided)" if raw_gap[-1] < 0 else ""}', flush=True)
            if bool(stop_eos(ids, None)[0]) or bool(stop_len(ids, None)[0]):
                break
            pos = torch.tensor([[past[0][0].shape[2]]])
            o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=past, position_ids=pos, use_cache=True, return_dict=True)
            row, past = o.logits[:, -1, :].float(), o.past_key_values
    rec = {f'{tag}.ids': np.array(toks, dtype=np.int64), f'{tag}.margin': np.array(margins), f'{tag}.raw_gap': np.array(raw_gap)}
    if keep_top:
        rec.update({f'{tag}.top16_ids': np.stack(top_i), f'{tag}.top16_logits': np.stack(top_v)})
    return rec, past0


def long_prompts(cfg, plan):
    """Text-only prompts of the LONG streams: seeded ids, the last one the walk's entry token (a CPU draw the GPU test repeats)."""
    g = torch.Generator().manual_seed(20261003)
    la, lb, lc = cfg['prompt_lens']
    entry_c = plan['chain_a'][cfg['long_c_entry'] - 1]            # the token after which walk A continues with chain_a[long_c_entry:]
    out = []
    for n, last in ((la, cfg['start_a']), (lb, plan['start_b']), (lc, entry_c)):
        ids = torch.randint(100, 60000, (n,), generator=g)
        ids[-1] = last
        out.append(ids)
    return out


def main_long():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    cfg = json.load(open(os.path.join(REF, 'InternVL', 'config.json')))
    P = synthetic.PEAKED_LONG
    dims = ModelDims.reduced(llm_layers=P['llm_layers'])
    plan = synthetic.peaked_plan(dims.vocab, P['start_a'], SEED, P)
    t0 = time.time()
    llm = build_peaked_llm(cfg, P['start_a'], dims, P)
    print(f'[{time.time() - t0:.0f}s] peaked LLM ({dims.llm_layers} layers) built', flush=True)
    prompts = long_prompts(P, plan)
    gold = {}
    expect = {'LA': plan['chain_a'][:1024], 'LB': plan['chain_b'], 'LC': plan['chain_a'][P['long_c_entry']:]}
    for tag, ids in zip(('LA', 'LB', 'LC'), prompts):
        with torch.no_grad():
            emb = llm.get_input_embeddings()(ids.reshape(1, -1))              # text only: generate_ocr's :1107 branch
        rec, _ = run_stream(llm, emb, 1.5, 1024, P['eos'], tag, t0, keep_top=False, quiet=True)
        got = rec[f'{tag}.ids'].tolist()
        print(f'{tag}: {len(got)} tokens, equal to the built walk: {got == expect[tag]}, margin min {rec[tag + ".margin"].min():.3f}', flush=True)
        gold[f'{tag}.ids'] = rec[f'{tag}.ids'].astype(np.int32)
        gold[f'{tag}.margin'] = rec[f'{tag}.margin'].astype(np.float16)
        gold[f'{tag}.penalty_decided'] = (rec[f'{tag}.raw_gap'] < 0).nonzero()[0].astype(np.int32)
        gold[f'{tag}.prompt'] = ids.numpy().astype(np.int32)
    meta = {'seed': SEED, 'peaked_long': {k: (list(v) if isinstance(v, tuple) else v) for k, v in P.items()}, 'penalty': 1.5, 'max_new_tokens': 1024,
            'torch': torch.__version__, 'transformers': __import__('transformers').__version__}
    gold['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, 'peaked_long.npz')
    np.savez_compressed(path, **gold)
    print('wrote', path, os.path.getsize(path), 'bytes', f'{time.time() - t0:.0f}s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--calibrate', action='store_true')
    ap.add_argument('--long', action='store_true')
    args = ap.parse_args()
    if args.long:
        return main_long()
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    cfg = json.load(open(os.path.join(REF, 'InternVL', 'config.json')))
    dims = ModelDims.full()
    P = synthetic.PEAKED
    ids_a = torch.from_numpy(np.load(os.path.join(OUT, 'config1_full_depth.npz'))['input_ids']).reshape(1, -1)
    start_a = int(ids_a[0, -1])
    plan = synthetic.peaked_plan(dims.vocab, start_a, SEED)
    ids_b = ids_a.clone()
    ids_b[0, -1] = plan['start_b']
    if args.calibrate:
        ids_a, ids_b = ids_a[:, -96:], ids_b[:, -96:]
    t0 = time.time()
    llm = build_peaked_llm(cfg, start_a)
    print(f'[{time.time() - t0:.0f}s] peaked LLM built', flush=True)
    n_vit, n_ref = int((ids_a == 92546).sum()), int((ids_a == 92537).sum())
    vit, ref = synthetic.peaked_prompt_embeds(max(n_vit, 1), max(n_ref, 1), seed=SEED)

    def embed(ids):                                   # generate_ocr head, modeling_internvl_chat.py:1087-1105
        with torch.no_grad():
            emb = llm.get_input_embeddings()(ids)
            B, N, C = emb.shape
            emb = emb.reshape(B * N, C)
            flat = ids.reshape(B * N)
            sel = flat == 92546
            if int(sel.sum()):
                emb[sel] = vit[:int(sel.sum())].reshape(-1, C)
            sel = flat == 92537
            if int(sel.sum()):
                emb[sel] = ref[:int(sel.sum())].reshape(-1, C).to(emb.dtype)
            return emb.reshape(B, N, C)

    gold = {}
    if args.calibrate:
        rec, _ = run_stream(llm, embed(ids_a), 1.0, 6, P['eos'], 'A', t0)
        print('A ids', rec['A.ids'].tolist(), 'built', plan['chain_a'][:6])
        rec, past0 = run_stream(llm, embed(ids_b), 1.5, 12, P['eos'], 'B15', t0)
        print('B15 ids', rec['B15.ids'].tolist(), 'built', plan['chain_b'][:12])
        rec, _ = run_stream(llm, None, 1.0, 16, P['eos'], 'B10', t0, past0)
        print('B10 ids', rec['B10.ids'].tolist(), 'built', plan['loop_b'][:16])
        return
    rec, _ = run_stream(llm, embed(ids_a), 1.0, 1024, P['eos'], 'A', t0)
    gold.update(rec)
    rec, past0 = run_stream(llm, embed(ids_b), 1.5, 1024, P['eos'], 'B15', t0)
    gold.update(rec)
    rec, _ = run_stream(llm, None, 1.0, 40, P['eos'], 'B10', t0, past0)
    gold.update(rec)
    gold['input_ids_a'] = ids_a[0].numpy().astype(np.int64)
    gold['input_ids_b'] = ids_b[0].numpy().astype(np.int64)
    gold['built.chain_a'] = np.array(plan['chain_a'], dtype=np.int64)
    gold['built.chain_b'] = np.array(plan['chain_b'], dtype=np.int64)
    gold['built.loop_b'] = np.array(plan['loop_b'], dtype=np.int64)
    meta = {'seed': SEED, 'peaked': {k: (list(v) if isinstance(v, tuple) else v) for k, v in P.items()}, 'start_a': start_a,
            'streams': {'A': {'penalty': 1.0, 'max_new_tokens': 1024}, 'B15': {'penalty': 1.5, 'max_new_tokens': 1024},
                        'B10': {'penalty': 1.0, 'max_new_tokens': 40}},
            'torch': torch.__version__, 'transformers': __import__('transformers').__version__}
    gold['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, 'peaked_streams.npz')
    np.savez_compressed(path, **gold)
    for tag in ('A', 'B15', 'B10'):
        m = gold[f'{tag}.margin']
        print(f'{tag}: {len(m)} tokens, margin min {m.min():.3f} median {np.median(m):.3f}, >= 1.0 on {100 * (m >= 1.0).mean():.0f} %')
    print('wrote', path, os.path.getsize(path), 'bytes', f'{time.time() - t0:.0f}s')


if __name__ == '__main__':
    main()
