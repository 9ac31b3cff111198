"""End-to-end drop-in check on a real MI355X: InternVLChatModel.chat_ocr / chat on the HIP engine vs the same
pipeline composed from the CPU oracle (1-layer ViT / resampler / LLM at full width, a fake tokenizer, a synthetic
page with character boxes).  Checks the host mirror (prompt assembly, tile preprocessing, splice, greedy loop,
EOS handling, return values) together with the kernels."""
import numpy as np
import pytest
import torch
from PIL import Image

from callireader_amd.config import ModelDims
from callireader_amd import synthetic, preprocess
from callireader_amd.conversation import get_conv_template

pytestmark = pytest.mark.gpu

from chat_helpers import SPECIALS, FakeTokenizer, oracle_chat_ocr


@pytest.fixture(scope='module')
def setup():
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1, vocab=9000)
    sd = synthetic.make_state_dict(dims, seed=0)
    model = InternVLChatModel.from_state_dict(sd, dims, max_tokens=4096)
    model.aligned_token_id = SPECIALS['[UNUSED_TOKEN_140]']
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 255, (500, 640, 3), dtype=np.uint8))
    boxes = [(10, 20, 110, 140), (200, 50, 420, 300), (300, 310, 380, 480)]
    return dict(model=model, dims=dims, sd=sd, img=img, boxes=boxes, tok=FakeTokenizer())


def test_chat_ocr_matches_oracle_pipeline(setup):
    m, tok = setup['model'], setup['tok']
    gen = dict(num_beams=1, max_new_tokens=6, do_sample=False)
    ref_ids, ref_q, n_tiles = oracle_chat_ocr(setup['sd'], setup['dims'], setup['img'], setup['boxes'], tok, 'what?', 6, 1.5)
    resp, hist = m.chat_ocr(tok, None, setup['img'], 'what?', gen, use_p=True, repetition_penalty=1.5, return_history=True,
                            boxes=setup['boxes'])
    assert n_tiles == 13                                   # 640x500 -> 4x3 tiles + thumbnail
    assert hist == [(ref_q, resp)]
    assert resp == tok.batch_decode(ref_ids)[0].split('<|im_end|>')[0].strip()
    assert gen == dict(num_beams=1, max_new_tokens=6, do_sample=False)       # caller's dict is not mutated
    # the GPU tile preprocessing (default) and the PIL host path give the same answer
    m.gpu_preprocess = False
    resp_pil = m.chat_ocr(tok, None, setup['img'], 'what?', gen, use_p=True, repetition_penalty=1.5, boxes=setup['boxes'])
    m.gpu_preprocess = True
    assert resp_pil == resp
    # str return when return_history is False; use_p=False path (no pseudo-tokens)
    ref_ids2, _, _ = oracle_chat_ocr(setup['sd'], setup['dims'], setup['img'], setup['boxes'], tok, 'what?', 4, 1.0, use_p=False)
    r2 = m.chat_ocr(tok, None, setup['img'], 'what?', dict(gen, max_new_tokens=4), use_p=False, repetition_penalty=1.0)
    assert isinstance(r2, str) and r2 == tok.batch_decode(ref_ids2)[0].split('<|im_end|>')[0].strip()


def test_chat_ocr_second_turn_on_the_first_turns_history(setup):
    """The reference's choice / bilingual / intent tasks (evaluate.py:259-287, 324-342) ask for the transcription and then put the real
    question as a second turn with the first turn as history: the second prompt holds the pseudo-token ids inside the history's question
    only, its own '<image>' stays literal text (only the first one is expanded, :716-719), and the same pseudo tokens are spliced again."""
    m, tok = setup['model'], setup['tok']
    gen = dict(num_beams=1, max_new_tokens=5, do_sample=False)
    r1, hist = m.chat_ocr(tok, None, setup['img'], 'what?', gen, use_p=True, repetition_penalty=1.0, return_history=True, boxes=setup['boxes'])
    first = list(hist)
    ref_ids, ref_q, _ = oracle_chat_ocr(setup['sd'], setup['dims'], setup['img'], setup['boxes'], tok, 'who wrote it?', 5, 1.0, history=first)
    r2, hist2 = m.chat_ocr(tok, None, setup['img'], 'who wrote it?', gen, use_p=True, repetition_penalty=1.0, return_history=True,
                           boxes=setup['boxes'], history=hist)
    assert r2 == tok.batch_decode(ref_ids)[0].split('<|im_end|>')[0].strip()
    assert hist2 == first + [(ref_q, r2)] and ref_q == '<image>\nwho wrote it?'


def test_chat_plain_and_errors(setup):
    m, tok = setup['model'], setup['tok']
    px = preprocess.load_image(setup['img']).to(torch.bfloat16)
    r = m.chat(tok, px.cuda(), 'hello', dict(num_beams=1, max_new_tokens=3, do_sample=False))
    assert isinstance(r, str) and len(r.split()) <= 3
    with pytest.raises(AssertionError):
        m.chat(tok, px.cuda(), 'hello', dict(max_new_tokens=2), num_patches_list=[1])      # len(pixel_values) != sum(num_patches_list)
    with pytest.raises(FileNotFoundError):
        m.chat_ocr(tok, None, '/nonexistent.jpg', 'q', dict(max_new_tokens=2), boxes=setup['boxes'])
    with pytest.raises(NotImplementedError):
        m.chat_ocr(tok, object(), setup['img'], 'q', dict(max_new_tokens=2))               # no boxes, detector not callable
    with pytest.raises(NotImplementedError):
        m.chat(tok, px.cuda(), 'hello', dict(num_beams=4, max_new_tokens=2))
    # detector as a callable
    r = m.chat_ocr(tok, lambda img: setup['boxes'][:1], setup['img'], 'q', dict(max_new_tokens=2), repetition_penalty=1.0)
    assert isinstance(r, str)


def test_eos_stops_generation(setup):
    """Force EOS: make '<|im_end|>' the argmax by handing generate_ocr an eos id equal to the first greedy pick."""
    m, tok = setup['model'], setup['tok']
    px = preprocess.load_image(setup['img']).to(torch.bfloat16).cuda()
    m.img_context_token_id = SPECIALS['<IMG_CONTEXT>']
    t = get_conv_template('internlm2-chat'); t.append_message(t.roles[0], '<image>\nhi'); t.append_message(t.roles[1], None)
    query = t.get_prompt().replace('<image>', '<img>' + '<IMG_CONTEXT>' * 256 * px.shape[0] + '</img>', 1)
    ids = tok(query)['input_ids']
    free = m.generate_ocr(pixel_values=px, input_ids=ids, repetition_penalty=1.0, max_new_tokens=40, eos_token_id=None)
    assert free.shape == (1, 40)
    stop_at = 17
    eos = int(free[0, stop_at])
    first = free[0].tolist().index(eos)
    out = m.generate_ocr(pixel_values=px, input_ids=ids, repetition_penalty=1.0, max_new_tokens=40, eos_token_id=eos)
    assert out[0].tolist() == free[0, :first + 1].tolist()            # EOS included, nothing after it


def test_batch_chat_equals_single_chats(setup):
    m, tok = setup['model'], setup['tok']
    px = preprocess.load_image(setup['img']).to(torch.bfloat16).cuda()
    gen = dict(num_beams=1, max_new_tokens=5, do_sample=False)
    singles = [m.chat(tok, px, q, gen) for q in ('alpha', 'beta gamma')]
    both = m.batch_chat(tok, torch.cat([px, px]), ['alpha', 'beta gamma'], gen, num_patches_list=[px.shape[0], px.shape[0]])
    assert both == singles
