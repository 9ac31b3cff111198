"""Token-exact FREE-RUNNING greedy decode at full depth against the reference, on a real MI355X.

tests/golden/peaked_streams.npz (scripts/make_golden_peaked.py) holds what the reference's own InternLM2ForCausalLM (32 layers,
vocabulary 92 553, eager, bf16) generates, token by token from its own picks, on the margin-controlled checkpoint of
callireader_amd.synthetic (`iter_peaked_llm`: the seed-0 architecture and tensors with wo / w2 scaled by 2^-9 and an LM head that
plants a walk through the vocabulary), from the config-1 prompt (3 158 ids of the reference's template and tokenizer for
examples/0.jpg, both masked overwrites done), driven the way InternVL/modeling_internvl_chat.py:1111-1120 drives
`language_model.generate` with the installed transformers' RepetitionPenaltyLogitsProcessor / EosTokenCriteria /
MaxLengthCriteria:
  A    repetition_penalty 1.0: 72 tokens, the last one EOS 92542;
  B15  repetition_penalty 1.5: 80 tokens ending in EOS; at 7 steps the raw arg-max is an id generated earlier and only the
       penalty makes the reference walk on;
  B10  the B prompt with repetition_penalty 1.0 and max_new_tokens 40: loops, stops on the length.
The reference's top-2 margin is >= 1.8 at every step, so there is NO near-tie escape here: the HIP path, through the drop-in
`InternVLChatModel.generate_ocr`, must return exactly the reference's ids -- same length, same EOS.  The raw logits of every
step are compared as well (top-16 of the reference).  Results: profiles/round4/peaked_streams.json (round 3's run: profiles/round3/).
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'peaked_streams.npz')
RESULTS = {}


def _dump():
    for d in (os.path.join(ROOT, 'profiles', 'round4'), os.path.join(ROOT, 'gpurun_out')):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, 'peaked_streams.json'), 'w') as f:
                json.dump(RESULTS, f, indent=1)
        except OSError:
            pass


@pytest.fixture(scope='module')
def gold():
    g = np.load(GOLD)
    return g, json.loads(bytes(g['meta']).decode())


@pytest.fixture(scope='module')
def model(gold):
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    g, meta = gold
    dims = ModelDims.full()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    m = InternVLChatModel(dims, max_tokens=4224, max_pages=2)
    t0 = time.time()
    for k, v in synthetic.iter_peaked_llm(dims, meta['start_a'], seed=meta['seed']):      # CPU draw == the golden script's weights
        m.engine.load_weight(k, v)
    m._finish()
    m.img_context_token_id = 92546
    RESULTS['weights_s'] = round(time.time() - t0, 1)
    n_vit, n_ref = int((g['input_ids_a'] == 92546).sum()), int((g['input_ids_a'] == 92537).sum())
    m._peaked_embeds = synthetic.peaked_prompt_embeds(n_vit, n_ref, seed=meta['seed'])
    yield m
    m.engine.close()


def _run(model, ids, penalty, max_new):
    vit, ref = model._peaked_embeds
    dummy_px = torch.zeros(1)            # generate_ocr takes the features from visual_features= when they are given (:1083-1086)
    out = model.generate_ocr(pixel_values=dummy_px, input_ids=torch.from_numpy(ids).reshape(1, -1), visual_features=vit.cuda(),
                             reference_embeds=ref.cuda(), repetition_penalty=penalty, num_beams=1, max_new_tokens=max_new,
                             do_sample=False, eos_token_id=synthetic.PEAKED['eos'])
    return out[0].tolist()


@pytest.mark.parametrize('tag', ['A', 'B15', 'B10'])
def test_free_running_ids_equal_the_reference(gold, model, tag):
    g, meta = gold
    cfg = meta['streams'][tag]
    ids = g['input_ids_a'] if tag == 'A' else g['input_ids_b']
    ref_ids = g[f'{tag}.ids'].tolist()
    got = _run(model, ids, cfg['penalty'], cfg['max_new_tokens'])
    first = next((i for i, (a, b) in enumerate(zip(got, ref_ids)) if a != b), None)
    RESULTS[tag] = {'penalty': cfg['penalty'], 'max_new_tokens': cfg['max_new_tokens'], 'reference_tokens': len(ref_ids), 'hip_tokens': len(got),
                    'ids_equal': got == ref_ids, 'first_difference': first, 'ends_with_eos': got[-1] == synthetic.PEAKED['eos'],
                    'reference_margin_min': float(g[f'{tag}.margin'].min()), 'reference_margin_median': float(np.median(g[f'{tag}.margin'])),
                    'steps_decided_by_the_penalty': int((g[f'{tag}.raw_gap'] < 0).sum()), 'hip_ids': got}
    _dump()
    print(f'peaked stream {tag}:', json.dumps({k: v for k, v in RESULTS[tag].items() if k != 'hip_ids'}))
    assert got == ref_ids, (tag, first)


def test_free_running_logits_and_batched_decode(gold, model):
    """The same streams step by step through the engine: raw logits of every free-running step against the reference's top-16, and
    streams A and B10 decoded TOGETHER as one batch (cr_llm_decode over two sequences) give their single-stream ids."""
    g, meta = gold
    eng = model.engine
    vit, ref = model._peaked_embeds
    out = {}
    kv = eng.kv_alloc(2, 3328)
    embs = {}
    for tag, ids in (('A', g['input_ids_a']), ('B', g['input_ids_b'])):
        embs[tag] = eng.embed_splice(torch.from_numpy(ids), vit.cuda(), ref.cuda())
    for tag, key, penalty in (('A', 'A', 1.0), ('B15', 'B', 1.5)):
        ref_ids = g[f'{tag}.ids'].tolist()
        top_i, top_v = torch.from_numpy(g[f'{tag}.top16_ids']), torch.from_numpy(g[f'{tag}.top16_logits'])
        kv.reset()
        lg = eng.prefill(kv, 0, embs[key], penalty=penalty, want_logits=True)
        worst, worst_top1 = 0.0, 0.0
        for t in range(len(ref_ids)):
            row = lg.float().cpu().reshape(-1)
            d = (row[top_i[t]] - top_v[t]).abs()
            worst, worst_top1 = max(worst, float(d.max())), max(worst_top1, float(d[0]))
            assert kv.generated(0)[t] == ref_ids[t], (tag, t)
            if t + 1 < len(ref_ids):
                lg = eng.decode(kv, [0], penalty=penalty, want_logits=True)
        out[tag] = {'steps': len(ref_ids), 'max_abs_diff_on_reference_top16': worst, 'max_abs_diff_on_reference_top1': worst_top1,
                    'reference_top1_logit_range': [float(top_v[:, 0].min()), float(top_v[:, 0].max())]}
        # bf16 logits at magnitude 8-16 step by 0.0625; the yardstick run (random weights, tests/test_gpu_full_depth.py) allows 0.49
        assert worst <= 0.25, out[tag]
    kv.reset()
    eng.prefill_batch(kv, [1, 0], [embs['A'], embs['B']], penalty=1.0)
    for _ in range(39):
        eng.decode(kv, [0, 1], penalty=1.0)
    out['batched'] = {'A_first_40_equal': kv.generated(1)[:40] == g['A.ids'].tolist()[:40], 'B10_equal': kv.generated(0)[:40] == g['B10.ids'].tolist()}
    kv.free()
    RESULTS['logits'] = out
    _dump()
    print('peaked streams, logits and batch:', json.dumps(out))
    assert out['batched']['A_first_40_equal'] and out['batched']['B10_equal'], out['batched']


def test_fp8_options_on_the_peaked_checkpoint_are_recorded(gold, model):
    """The fp8 switches are throughput options, OFF by default; this records what they do to a stream whose reference margins are wide
    (>= 2.6 at |logit| ~ 12): tokens of stream A equal to the reference's, first difference, per switch.  No equality is asserted -- the
    record (profiles/round4/peaked_streams.json (round 3's run: profiles/round3/), key fp8) is the honest answer to "does fp8 keep the tokens on these weights"."""
    g, meta = gold
    ref_ids = g['A.ids'].tolist()
    eng = model.engine
    out = {}
    for name, on, off in (('fp8_mfma_level1', lambda: eng.enable_fp8_mfma(True, level=1), lambda: eng.enable_fp8_mfma(False)),
                          ('fp8_mfma_level2', lambda: eng.enable_fp8_mfma(True, level=2), lambda: eng.enable_fp8_mfma(False)),
                          ('fp8_decode', lambda: eng.enable_fp8_decode(True), lambda: eng.enable_fp8_decode(False))):
        on()
        try:
            got = _run(model, g['input_ids_a'], 1.0, len(ref_ids) + 8)
        finally:
            off()
        first = next((i for i, (a, b) in enumerate(zip(got, ref_ids)) if a != b), None)
        out[name] = {'tokens': len(got), 'equal_to_reference': got == ref_ids, 'first_difference': first,
                     'tokens_equal_before_it': first if first is not None else min(len(got), len(ref_ids))}
    assert _run(model, g['input_ids_a'], 1.0, 1024) == ref_ids          # switched off again: the bf16 stream is back
    RESULTS['fp8'] = out
    _dump()
    print('peaked stream A under the fp8 options:', json.dumps(out))
