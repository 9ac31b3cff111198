"""BASELINE config 1 shape on the GPU: the reference's own example page (tests/golden/example0.jpg, 788x2000) with its
96 labelled character boxes -> 11 page tiles + 96 character tiles -> prompt of ~3.2k tokens -> greedy decode, HIP engine
vs the oracle-composed pipeline (1 layer per stage at full width, seeded weights, fake tokenizer).  No real weights
exist offline, so this checks plumbing and token parity, not OCR quality."""
import json
import os

import pytest
import torch
from PIL import Image

from callireader_amd.config import ModelDims
from callireader_amd import synthetic, preprocess
from chat_helpers import SPECIALS, FakeTokenizer, oracle_chat_ocr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_example_page_end_to_end():
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    img = Image.open(os.path.join(GOLD, 'example0.jpg')).convert('RGB')
    boxes = preprocess.boxes_from_labelme(json.load(open(os.path.join(GOLD, 'example0_boxes.json'))))
    assert img.size == (788, 2000) and len(boxes) == 96
    dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1, vocab=9000)
    sd = synthetic.make_state_dict(dims, seed=0)
    m = InternVLChatModel.from_state_dict(sd, dims, max_tokens=4096)
    m.aligned_token_id = SPECIALS['[UNUSED_TOKEN_140]']
    tok = FakeTokenizer()
    q = '这幅书法作品内容是什么？'                                       # inference.py:69 default prompt
    ref_ids, ref_q, n_tiles = oracle_chat_ocr(sd, dims, img, boxes, tok, q, 8, 1.0)
    assert n_tiles == 11                                                 # (2,5) grid + thumbnail
    resp, hist = m.chat_ocr(tok, None, img, q, dict(num_beams=1, max_new_tokens=8, do_sample=False), use_p=True,
                            repetition_penalty=1.0, return_history=True, boxes=boxes)
    assert hist[0][0] == ref_q and ref_q.count('[UNUSED_TOKEN_140]') == 96 * 3
    assert m.kv().length(0) >= 11 * 256 + 96 * 3
    assert resp == tok.batch_decode(ref_ids)[0].split('<|im_end|>')[0].strip()
