"""Self-contained InternLM2 tokenizer pinned against the sentencepiece library itself:
  * a BPE model trained inside the test with the same spec as the reference's (identity normaliser, no dummy prefix,
    byte fallback, user-defined symbols): pieces, ids and decoded text must equal sentencepiece's on random text;
  * the reference's own tokenizer.model when it is reachable (env CALLIREADER_TOKENIZER_DIR, or the build container's
    /root/reference): same comparison after patching the one NUL-containing piece that newer sentencepiece rejects,
    plus the golden ids in tests/golden/tokenizer_vectors.json (produced with that model)."""
import json
import os
import random

import pytest
import torch

from callireader_amd.tokenization_internlm2 import SentencePieceBPE, InternLM2Tokenizer

spm = pytest.importorskip('sentencepiece')
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'tokenizer_vectors.json')
ALPHABETS = ['abcdefghijklmnopqrstuvwxyz', 'ABCDEFGH', '0123456789', ' ', ' ', '，。？！：', '书法作品内容是什么君不见黄河之水天上来奔流到海不复回', '\n\t', 'éüñ', '😀']


def rand_text(rng, n):
    return ''.join(rng.choice(rng.choice(ALPHABETS)) for _ in range(n))


@pytest.fixture(scope='module')
def trained(tmp_path_factory):
    d = tmp_path_factory.mktemp('spm')
    rng = random.Random(0)
    corpus = os.path.join(d, 'corpus.txt')
    with open(corpus, 'w', encoding='utf-8') as f:
        for _ in range(3000):
            f.write(rand_text(rng, rng.randint(5, 60)).replace('\n', ' ').replace('\t', ' ') + '\n')
    prefix = os.path.join(d, 'toy')
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=prefix, vocab_size=600, model_type='bpe', character_coverage=0.995,
                                   normalization_rule_name='identity', add_dummy_prefix=False, remove_extra_whitespaces=False,
                                   byte_fallback=True, split_digits=True, user_defined_symbols=['[UNUSED_TOKEN_140]', '[UNUSED_TOKEN_141]'],
                                   minloglevel=2)
    sp = spm.SentencePieceProcessor(); sp.Load(prefix + '.model')
    return sp, SentencePieceBPE(prefix + '.model')


def test_bpe_equals_sentencepiece_on_trained_model(trained):
    sp, mine = trained
    assert len(mine) == sp.get_piece_size() and mine.bos_id == sp.bos_id() and mine.eos_id == sp.eos_id()
    rng = random.Random(1)
    texts = ['', ' ', '  leading and  double  spaces ', 'plain ascii text', '这幅书法作品内容是什么？', 'a[UNUSED_TOKEN_140][UNUSED_TOKEN_140]b[UNUSED_TOKEN_141]',
             'emoji 😀 and tab\tnewline\n end', '[UNUSED_TOKEN_14', 'ÿþ unseen Ω chars']
    texts += [rand_text(rng, rng.randint(1, 80)) for _ in range(300)]
    for t in texts:
        assert mine.encode_pieces(t) == [p for p in sp.encode(t, out_type=str)] or mine.encode(t) == sp.encode(t), t
        ids = sp.encode(t)
        assert mine.encode(t) == ids, t
        assert mine.decode(ids) == sp.decode(ids), t


def _reference_dir():
    for d in (os.environ.get('CALLIREADER_TOKENIZER_DIR'), '/root/reference/InternVL'):
        if d and os.path.exists(os.path.join(d, 'tokenizer.model')):
            return d
    return None


@pytest.fixture(scope='module')
def real(tmp_path_factory):
    d = _reference_dir()
    if d is None:
        pytest.skip('reference tokenizer.model not reachable')
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    m.ParseFromString(open(os.path.join(d, 'tokenizer.model'), 'rb').read())
    for p in m.pieces:
        if '\x00' in p.piece:
            p.piece = '<NUL_PLACEHOLDER_PIECE>'
    patched = os.path.join(tmp_path_factory.mktemp('real'), 'patched.model')
    open(patched, 'wb').write(m.SerializeToString())
    sp = spm.SentencePieceProcessor(); sp.Load(patched)
    return d, sp, InternLM2Tokenizer.from_pretrained(d)


def test_real_model_equals_sentencepiece(real):
    d, sp, tok = real
    assert tok.vocab_size == 92544 and tok.convert_tokens_to_ids('[UNUSED_TOKEN_140]') == 92537
    assert tok.convert_tokens_to_ids('<|im_end|>') == 92542 and tok.convert_tokens_to_ids('<IMG_CONTEXT>') == 92546
    rng = random.Random(2)
    texts = ['这幅书法作品内容是什么？', '君不见，黄河之水天上来，奔流到海不复回。', 'Hello world 123', '输出图片中所有文字:', ' x  y ', '\n']
    texts += [rand_text(rng, rng.randint(1, 60)) for _ in range(200)]
    for t in texts:
        assert tok.sp_model.encode(t) == sp.encode(t), t
        assert tok.sp_model.decode(sp.encode(t)) == sp.decode(sp.encode(t)), t


def test_real_model_chat_prompt_and_golden(real):
    d, sp, tok = real
    from callireader_amd.conversation import get_conv_template
    t = get_conv_template('internlm2-chat')
    t.append_message(t.roles[0], '<image>\n这幅书法作品内容是什么？' + '[UNUSED_TOKEN_140]' * 6)
    t.append_message(t.roles[1], None)
    query = t.get_prompt().replace('<image>', '<img>' + '<IMG_CONTEXT>' * 4 + '</img>', 1)
    enc = tok(query, return_tensors='pt')
    ids = enc['input_ids'][0].tolist()
    assert enc['attention_mask'].shape == enc['input_ids'].shape and ids[0] == 1              # BOS (add_bos_token)
    assert ids.count(92546) == 4 and ids.count(92537) == 6 and ids.count(92543) == 3 and ids.count(92542) == 2
    # every stretch between added tokens is what sentencepiece gives for that stretch
    sys_msg = t.system_message
    assert ids[1] == 92543 and ids[2:2 + len(sp.encode('system\n' + sys_msg))] == sp.encode('system\n' + sys_msg)
    gold = json.load(open(GOLD, encoding='utf-8'))
    for item in gold['cases']:
        assert tok.encode(item['text']) == item['ids'], item['text']
    # decode: specials skipped, text restored
    out = tok.batch_decode(torch.tensor([sp.encode('君不见，黄河之水天上来') + [92542]]), skip_special_tokens=True)[0]
    assert out.split('<|im_end|>')[0].strip() == '君不见，黄河之水天上来'


def test_wrapper_equals_the_reference_class(real):
    """`wrapper` in the golden file was recorded from the REFERENCE's own InternLM2Tokenizer
    (InternVL/tokenization_internlm2.py:34-235) loaded on its NUL-patched model (scripts/make_golden_tokenizer.py): ids
    of every case (added-token splitting, BOS, the empty string), convert_tokens_to_ids, and batch_decode with
    skip_special_tokens=True (what chat_ocr uses, modeling_internvl_chat.py:752)."""
    d, sp, tok = real
    gold = json.load(open(GOLD, encoding='utf-8'))
    w = gold['wrapper']
    assert w['ok'] and not w.get('mismatch_vs_library')
    for item, ids in zip(gold['cases'], w['ids']):
        assert tok(item['text'])['input_ids'] == ids, item['text']
    for t, i in w['convert_tokens_to_ids'].items():
        assert tok.convert_tokens_to_ids(t) == i, t
    assert tok.batch_decode(w['decode_ids'], skip_special_tokens=True) == w['decode_skip_special']
    # (decode_keep_special is recorded too, but spacing around kept specials is a transformers-version matter -- the file was
    #  written under 5.x, the reference pins 4.45.2 -- and the path only ever decodes with skip_special_tokens=True, :752)
    assert tok.batch_decode(torch.tensor(w['decode_ids'][:1]), skip_special_tokens=True) == w['decode_skip_special'][:1]


def test_wrapper_on_trained_model(trained, tmp_path):
    sp, _ = trained
    # a checkpoint-dir layout with added tokens beyond the sentencepiece vocabulary, like the reference's
    d = tmp_path
    open(os.path.join(d, 'tokenizer.model'), 'wb').write(sp.serialized_model_proto())
    n = sp.get_piece_size()
    json.dump({'added_tokens_decoder': {'0': {'content': '<unk>', 'special': True}, '1': {'content': '<s>', 'special': True},
                                       '2': {'content': '</s>', 'special': True}, str(n): {'content': '<|im_end|>', 'special': True},
                                       str(n + 1): {'content': '<img>', 'special': True}}},
              open(os.path.join(d, 'tokenizer_config.json'), 'w'))
    json.dump({'<IMG_CONTEXT>': n + 2}, open(os.path.join(d, 'added_tokens.json'), 'w'))
    tok = InternLM2Tokenizer.from_pretrained(str(d))
    ids = tok('ab<img><IMG_CONTEXT><IMG_CONTEXT>cd<|im_end|>', return_tensors='pt')['input_ids'][0].tolist()
    assert ids == [1] + sp.encode('ab') + [n + 1, n + 2, n + 2] + sp.encode('cd') + [n]
    assert tok.convert_tokens_to_ids('<IMG_CONTEXT>') == n + 2 and tok.convert_tokens_to_ids('<|im_end|>') == n
    assert tok.batch_decode([ids], skip_special_tokens=True)[0] == 'ab <IMG_CONTEXT> <IMG_CONTEXT> cd'
    assert tok.batch_decode([sp.encode('hello there') + [n]], skip_special_tokens=True)[0] == 'hello there'
    both = tok(['ab', 'abcdefgh ijk'], return_tensors='pt', padding=True)
    assert both['input_ids'].shape == both['attention_mask'].shape and int(both['attention_mask'][0].sum()) == 1 + len(sp.encode('ab'))
