"""A LONG free-running reference stream, id for id (round-5 verdict, item 5): tests/golden/peaked_long.npz holds what the reference's own InternLM2ForCausalLM
(2 layers, full width, vocabulary 92 553, eager, bf16; scripts/make_golden_peaked.py --long) generates on the margin-controlled checkpoint
`synthetic.PEAKED_LONG`, driven as InternVL/modeling_internvl_chat.py:1111-1120 drives `language_model.generate` with repetition_penalty 1.5 (chat_ocr's default, :652)
and max_new_tokens 1024 (inference.py:92-96), by the installed transformers' RepetitionPenaltyLogitsProcessor / EosTokenCriteria / MaxLengthCriteria:
  LA  1 024 tokens, stopped by the length (no EOS on the way): the cache grows 1 024 past a 333-token prompt, through four 256-key split boundaries;
  LB  EOS as its 85th token (85 = 5 mod 16: between two of the engine's EOS checks), eight steps at which only the penalty makes the reference walk on;
  LC  EOS as its 397th token (13 mod 16), 200-token prompt.
The three pages go through `generate_pages` as ONE batch (different prompt lengths, different stopping steps, check_every = 16) and through the two-batch pipeline;
margins >= 4.3 at every step: no near-tie escape, the ids must be the reference's."""
import json
import os

import numpy as np
import pytest
import torch

from callireader_amd import synthetic
from callireader_amd.config import ModelDims

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'peaked_long.npz')


@pytest.fixture(scope='module')
def setup():
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    g = np.load(GOLD)
    meta = json.loads(bytes(g['meta']).decode())
    P = synthetic.PEAKED_LONG
    dims = ModelDims.reduced(llm_layers=P['llm_layers'])
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    m = InternVLChatModel(dims, max_tokens=1408, max_pages=3)
    for k, v in synthetic.iter_peaked_llm(dims, P['start_a'], seed=meta['seed'], cfg=P):      # CPU draw == the golden script's weights
        m.engine.load_weight(k, v)
    m._finish()
    m.img_context_token_id = 92546                    # what chat_ocr sets from the tokenizer before it calls generate_ocr (:703)
    yield m, g, meta
    m.engine.close()


def test_three_pages_of_different_lengths_in_one_batch_equal_the_reference(setup):
    m, g, meta = setup
    eos = synthetic.PEAKED_LONG['eos']
    tags = ('LA', 'LB', 'LC')
    embeds = [m.engine.embed_splice(torch.from_numpy(g[f'{t}.prompt'].astype(np.int64))) for t in tags]
    want = [g[f'{t}.ids'].tolist() for t in tags]
    assert [len(w) for w in want] == [1024, 85, 397] and want[1][-1] == eos and want[2][-1] == eos and eos not in want[0]
    outs = m.generate_pages(embeds, max_new_tokens=meta['max_new_tokens'], eos_token_id=eos, repetition_penalty=meta['penalty'])
    for t, o, w in zip(tags, outs, want):
        first = next((i for i, (a, b) in enumerate(zip(o, w)) if a != b), None)
        assert o == w, (t, len(o), len(w), first)
    # the same pages in another order and as single pages: a page's ids do not depend on its batch or its slot
    outs2 = m.generate_pages([embeds[2], embeds[0], embeds[1]], max_new_tokens=meta['max_new_tokens'], eos_token_id=eos, repetition_penalty=meta['penalty'])
    assert outs2 == [want[2], want[0], want[1]]
    assert m.generate_pages([embeds[1]], max_new_tokens=meta['max_new_tokens'], eos_token_id=eos, repetition_penalty=meta['penalty']) == [want[1]]
    # ... and through the drop-in's one-page entry point with a shorter budget: the stream stops on the length, the ids are the reference's first 300
    out = m.generate_ocr(input_ids=torch.from_numpy(g['LA.prompt'].astype(np.int64)).reshape(1, -1), repetition_penalty=meta['penalty'], num_beams=1,
                         max_new_tokens=300, do_sample=False, eos_token_id=eos)
    assert out[0].tolist() == want[0][:300]


def test_two_batches_in_flight_give_the_same_long_streams(setup):
    """PagePipeline (chat_ocr_stream's engine): batch 1 = (LA, LC), batch 2 = (LB, LA); each batch decodes on the side stream while the next is prefilled."""
    m, g, meta = setup
    eos = synthetic.PEAKED_LONG['eos']
    e = {t: m.engine.embed_splice(torch.from_numpy(g[f'{t}.prompt'].astype(np.int64))) for t in ('LA', 'LB', 'LC')}
    w = {t: g[f'{t}.ids'].tolist() for t in ('LA', 'LB', 'LC')}
    pipe = m.page_pipeline(max_new_tokens=meta['max_new_tokens'], eos_token_id=eos, repetition_penalty=meta['penalty'])
    try:
        assert pipe.start([e['LA'], e['LC']]) is None
        first = pipe.start([e['LB'], e['LA']])
        second = pipe.finish()
    finally:
        pipe.close()
    assert first == [w['LA'], w['LC']] and second == [w['LB'], w['LA']]
