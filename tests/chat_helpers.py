"""Shared pieces of the end-to-end chat tests: a fake tokenizer and the same pipeline composed from the CPU oracle."""
import numpy as np
import torch
from PIL import Image

from callireader_amd import preprocess
from callireader_amd.conversation import get_conv_template

SPECIALS = {'<IMG_CONTEXT>': 8990, '[UNUSED_TOKEN_140]': 8991, '<|im_end|>': 8992, '<|im_start|>': 8993, '<img>': 8994, '</img>': 8995}


class FakeTokenizer:
    def _encode(self, text):
        ids, i = [], 0
        while i < len(text):
            for s, v in SPECIALS.items():
                if text.startswith(s, i):
                    ids.append(v); i += len(s); break
            else:
                ids.append(10 + ord(text[i]) % 7000); i += 1
        return ids

    def __call__(self, query, return_tensors='pt'):
        ids = torch.tensor([self._encode(query)], dtype=torch.long)
        return {'input_ids': ids, 'attention_mask': torch.ones_like(ids)}

    def convert_tokens_to_ids(self, tok):
        return SPECIALS[tok]

    def batch_decode(self, out, skip_special_tokens=True):
        return [' '.join(('<|im_end|>' if int(t) == 8992 else str(int(t))) for t in row) for row in out]


def oracle_chat_ocr(sd, dims, img, boxes, tok, question, max_new, penalty, use_p=True, drop_zero=False, history=None):
    from oracle import vision, calli_align, generate
    with torch.no_grad():
        page_px = preprocess.load_image(img).to(torch.bfloat16)
        q = '<image>\n' + question
        out_tokens = None
        if use_p:
            arr = np.array(img)
            tiles = torch.cat([preprocess.load_image_2(Image.fromarray(arr[y1:y2, x1:x2])).to(torch.bfloat16) for x1, y1, x2, y2 in boxes])
            feats = vision.extract_feature(sd, tiles, dims.vit_layers)
            rs = calli_align.resampler_forward(sd, feats, dims.rs_depth)
            idx = calli_align.vq_cos_sim(sd['normed_emb.weight'], rs)
            out_tokens, _ = calli_align.denormalise(rs, idx.reshape(rs.shape[0], 3), sd['normed_emb.weight'], sd['calli.mu'], sd['calli.sigma'], drop_zero=drop_zero)
            if history is None:                                   # a later turn carries the pseudo-token ids in its history only (:698-699)
                q = q + '[UNUSED_TOKEN_140]' * out_tokens.shape[0]
        t = get_conv_template('internlm2-chat')
        for old_q, old_a in (history or []):                      # :711-713
            t.append_message(t.roles[0], old_q); t.append_message(t.roles[1], old_a)
        t.append_message(t.roles[0], q); t.append_message(t.roles[1], None)
        query = t.get_prompt().replace('<image>', '<img>' + '<IMG_CONTEXT>' * 256 * page_px.shape[0] + '</img>', 1)
        ids = tok(query)['input_ids']
        vit = vision.extract_feature(sd, page_px, dims.vit_layers)
        emb = generate.splice_embeddings(sd, ids, vit, out_tokens, SPECIALS['<IMG_CONTEXT>'], SPECIALS['[UNUSED_TOKEN_140]'])
        out = generate.greedy_generate(sd, dims.llm_layers, emb, max_new_tokens=max_new, eos_token_id=SPECIALS['<|im_end|>'],
                                       repetition_penalty=penalty)
    return out, q, page_px.shape[0]


