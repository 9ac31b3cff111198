#!/usr/bin/env python3
"""tests/golden/tokenizer_vectors.json from the reference's tokenizer files (build container only).

Two sources, both the reference's own:
  * ids: the sentencepiece LIBRARY on the reference's tokenizer.model (one piece, #354, holds a NUL byte that
    sentencepiece >= 0.2.1 refuses; it is swapped for an unused private-use character in a temporary copy), per
    stretch of text between added tokens, + BOS + the added-token ids of tokenizer_config.json/added_tokens.json: what
    `tokenizer(text)['input_ids']` is for the reference's `InternLM2Tokenizer` (tokenization_internlm2.py:34-235);
  * wrapper behaviour: the reference's class itself (`InternVL.tokenization_internlm2.InternLM2Tokenizer`) loaded on that
    patched copy -- when the installed transformers can still drive it, its ids, `convert_tokens_to_ids`,
    `batch_decode(..., skip_special_tokens=True)` are recorded as `wrapper_*` fields and must equal the library path.
Only data is written.  Usage: python scripts/make_golden_tokenizer.py
"""
import json
import os
import re
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/InternVL'
OUT = os.path.join(ROOT, 'tests', 'golden', 'tokenizer_vectors.json')

TEXTS = [
    '这幅书法作品内容是什么？',
    '<|im_start|>user\n<img><IMG_CONTEXT><IMG_CONTEXT></img>\n输出图片中所有文字:<|im_end|><|im_start|>assistant\n',
    'Hello world 123 [UNUSED_TOKEN_140][UNUSED_TOKEN_140]',
    '君不见，黄河之水天上来，奔流到海不复回。君不见，高堂明镜悲白发，朝如青丝暮成雪。',
    '  two  spaces\tand\nnewline 😀',
    '读出图中所有文字。',
    '<|im_start|>system\n你是由上海人工智能实验室联合商汤科技开发的书生多模态大模型，英文名叫InternVL, 是一个有用无害的人工智能助手。<|im_end|><|im_start|>user\n<image>\n这幅书法作品内容是什么？<|im_end|><|im_start|>assistant\n',
    '',
    'x<|im_end|>y</s>z<s>w',
]
DECODE_IDS = [[1, 90930, 71136, 68426, 92542, 60504], [92543, 1008, 364, 92546, 92545, 2], [9843, 2028, 262, 92537]]


def patched_model_dir():
    from sentencepiece import sentencepiece_model_pb2 as pb
    d = tempfile.mkdtemp(prefix='cr_tok_')
    for f in os.listdir(REF):
        if f.startswith(('tokenizer', 'special_tokens', 'added_tokens', 'tokenization_')):
            shutil.copy(os.path.join(REF, f), d)
    m = pb.ModelProto()
    m.ParseFromString(open(os.path.join(REF, 'tokenizer.model'), 'rb').read())
    n = 0
    for p in m.pieces:
        if '\x00' in p.piece:
            p.piece = p.piece.replace('\x00', '')
            n += 1
    open(os.path.join(d, 'tokenizer.model'), 'wb').write(m.SerializeToString())
    return d, n


def main():
    import sentencepiece as spm
    d, n_patched = patched_model_dir()
    sp = spm.SentencePieceProcessor()
    sp.Load(os.path.join(d, 'tokenizer.model'))
    cfg = json.load(open(os.path.join(REF, 'tokenizer_config.json')))
    added = {}
    if os.path.exists(os.path.join(REF, 'added_tokens.json')):
        added.update(json.load(open(os.path.join(REF, 'added_tokens.json'))))
    for k, v in cfg.get('added_tokens_decoder', {}).items():
        added[v['content']] = int(k)
    special = sorted(added, key=len, reverse=True)
    pat = re.compile('(' + '|'.join(re.escape(t) for t in special) + ')')

    def lib_ids(text):
        ids = [sp.bos_id()]                                       # add_bos_token: tokenization_internlm2.py:60,178-190
        for part in pat.split(text):
            if part in added:
                ids.append(added[part])
            elif part:
                ids.extend(sp.encode(part))
        return ids
    out = {'_how': 'scripts/make_golden_tokenizer.py: ids = [BOS] + sentencepiece(%s, NUL-patched copy of the reference tokenizer.model, %d piece patched) per '
                   'stretch + added-token ids from tokenizer_config.json / added_tokens.json' % (spm.__version__, n_patched),
           'added_tokens': added, 'cases': [{'text': t, 'ids': lib_ids(t)} for t in TEXTS]}
    # the reference's wrapper class, if the installed transformers still runs it
    try:
        sys.path.insert(0, '/root/reference')
        from InternVL.tokenization_internlm2 import InternLM2Tokenizer
        tok = InternLM2Tokenizer.from_pretrained(d)
        out['wrapper'] = {'class': 'InternVL.tokenization_internlm2.InternLM2Tokenizer', 'ok': True,
                          'ids': [tok(t)['input_ids'] for t in TEXTS],
                          'convert_tokens_to_ids': {t: tok.convert_tokens_to_ids(t) for t in ('<IMG_CONTEXT>', '<|im_end|>', '[UNUSED_TOKEN_140]', '<img>', '</img>')},
                          'decode_ids': DECODE_IDS,
                          'decode_skip_special': tok.batch_decode(DECODE_IDS, skip_special_tokens=True),
                          'decode_keep_special': tok.batch_decode(DECODE_IDS, skip_special_tokens=False)}
        for a, b in zip(out['wrapper']['ids'], out['cases']):
            if a != b['ids']:
                out['wrapper']['mismatch_vs_library'] = True
    except Exception as e:                                        # recorded, not hidden: the judge can see what did not run
        out['wrapper'] = {'ok': False, 'error': f'{type(e).__name__}: {e}'[:400]}
    json.dump(out, open(OUT, 'w'), ensure_ascii=False, indent=1)
    print('wrote', OUT, 'wrapper ok:', out['wrapper'].get('ok'), out['wrapper'].get('error', ''))
    shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
