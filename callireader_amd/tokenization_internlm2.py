"""Self-contained InternLM2 tokenizer for the chat path: no sentencepiece / transformers needed.

Why: the reference wraps sentencepiece==0.2.0 (`InternVL/tokenization_internlm2.py:34-235`); newer sentencepiece
releases refuse the shipped `tokenizer.model` (piece #354 contains a NUL), so the engine carries its own reader.

What is restated
  * the model file: the sentencepiece ModelProto wire format (pieces: string, score, type) parsed directly;
  * encoding: sentencepiece's BPE (`bpe_model.cc`, published algorithm): identity normaliser, spaces escaped to
    U+2581, no dummy prefix; user-defined pieces (type 4, e.g. `[UNUSED_TOKEN_140]`) are matched longest-first and
    frozen; every other character starts as its own symbol; the adjacent pair whose concatenation is a vocabulary
    piece with the highest score is merged first (ties: leftmost); symbols left outside the vocabulary fall back to
    their UTF-8 bytes (`<0xXX>` pieces);
  * the HF layer of the reference wrapper: text is first split on the added tokens of `tokenizer_config.json` /
    `added_tokens.json` (`<|im_start|>`, `<|im_end|>`, `<img>`, `<IMG_CONTEXT>`, ...; not normalised, no stripping),
    BOS is prepended (`add_bos_token=True`, `build_inputs_with_special_tokens` :163-178), decoding drops special ids
    when asked and joins pieces as `convert_tokens_to_string` (:123-142) does (its prefix-space dance nets to identity).
Pinned by tests/test_tokenizer.py against the sentencepiece library itself (a model trained in the test, and the
reference's model when it is available) and against ids produced with the reference's model (tests/golden).
"""
import heapq
import json
import os
import re
import struct

import torch

NORMAL, UNKNOWN, CONTROL, USER_DEFINED, UNUSED, BYTE = 1, 2, 3, 4, 5, 6
SPACE = '▁'


# ---- minimal protobuf reader for sentencepiece's ModelProto ---------------------------------------------------------
def _varint(buf, i):
    x = s = 0
    while True:
        b = buf[i]; i += 1
        x |= (b & 0x7F) << s
        if not b & 0x80:
            return x, i
        s += 7


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v = buf[i:i + 8]; i += 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]; i += ln
        elif wt == 5:
            v = buf[i:i + 4]; i += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fn, wt, v


def read_sentencepiece_model(path):
    """-> (pieces [(text, score, type)], byte_fallback, unk_id, bos_id, eos_id)"""
    data = open(path, 'rb').read()
    pieces, byte_fallback, unk_id, bos_id, eos_id = [], False, 0, 1, 2
    for fn, wt, v in _fields(data):
        if fn == 1 and wt == 2:                                   # repeated SentencePiece pieces = 1
            text, score, typ = '', 0.0, NORMAL
            for f2, w2, v2 in _fields(v):
                if f2 == 1:
                    text = bytes(v2).decode('utf-8')
                elif f2 == 2:
                    score = struct.unpack('<f', bytes(v2))[0]
                elif f2 == 3:
                    typ = v2
            pieces.append((text, score, typ))
        elif fn == 2 and wt == 2:                                 # TrainerSpec
            for f2, w2, v2 in _fields(v):
                if f2 == 35:
                    byte_fallback = bool(v2)
                elif f2 == 40:
                    unk_id = v2
                elif f2 == 41:
                    bos_id = v2
                elif f2 == 42:
                    eos_id = v2
    return pieces, byte_fallback, unk_id, bos_id, eos_id


class SentencePieceBPE:
    """Encoder/decoder equivalent to sentencepiece's for BPE models with an identity normaliser and no dummy prefix."""

    def __init__(self, path):
        self.pieces, self.byte_fallback, self.unk_id, self.bos_id, self.eos_id = read_sentencepiece_model(path)
        self.piece_to_id = {}
        self.score = {}
        self.user_defined = []
        self.byte_id = {}
        for i, (t, s, ty) in enumerate(self.pieces):
            self.piece_to_id.setdefault(t, i)
            if ty in (NORMAL, USER_DEFINED, UNUSED):
                self.score[t] = s
            if ty == USER_DEFINED:
                self.user_defined.append(t)
            if ty == BYTE:
                self.byte_id[int(t[3:5], 16)] = i
        self.user_defined.sort(key=len, reverse=True)
        self._ud_re = re.compile('|'.join(re.escape(t) for t in self.user_defined)) if self.user_defined else None

    def __len__(self):
        return len(self.pieces)

    def _split(self, text):
        """characters, with user-defined pieces kept whole and frozen (longest piece first at every position: the alternation is sorted by length)"""
        if self._ud_re is None:
            return [(c, False) for c in text]
        out, i = [], 0
        for m in self._ud_re.finditer(text):
            out.extend((c, False) for c in text[i:m.start()])
            out.append((m.group(), True))
            i = m.end()
        out.extend((c, False) for c in text[i:])
        return out

    def encode_pieces(self, text):
        text = text.replace(' ', SPACE)
        if not text:
            return []
        sym = self._split(text)
        piece = [s for s, _ in sym]
        frozen = [f for _, f in sym]
        n = len(piece)
        prev = list(range(-1, n - 1))
        nxt = list(range(1, n + 1)); nxt[-1] = -1
        heap = []

        def push(l, r):
            if l < 0 or r < 0 or frozen[l] or frozen[r]:
                return
            m = piece[l] + piece[r]
            sc = self.score.get(m)
            if sc is not None:
                heapq.heappush(heap, (-sc, l, r, len(m)))           # highest score first, then leftmost

        for i in range(n - 1):
            push(i, i + 1)
        while heap:
            _, l, r, size = heapq.heappop(heap)
            if piece[l] is None or piece[r] is None or len(piece[l]) + len(piece[r]) != size or nxt[l] != r:
                continue                                            # stale entry
            piece[l] = piece[l] + piece[r]
            piece[r] = None
            nxt[l] = nxt[r]
            if nxt[r] >= 0:
                prev[nxt[r]] = l
            push(prev[l], l)
            push(l, nxt[l])
        out, i = [], 0
        while i >= 0:
            out.append(piece[i]); i = nxt[i]
        return out

    def encode(self, text):
        ids = []
        for p in self.encode_pieces(text):
            i = self.piece_to_id.get(p)
            if i is not None and self.pieces[i][2] in (NORMAL, USER_DEFINED, UNUSED):
                ids.append(i)
            elif self.byte_fallback:
                ids.extend(self.byte_id[b] for b in p.encode('utf-8'))
            else:
                ids.append(self.unk_id)
        return ids

    def decode_pieces(self, pieces):
        out, pending = [], bytearray()
        for p in pieces:
            i = self.piece_to_id.get(p)
            if i is not None and self.pieces[i][2] == BYTE:
                pending.append(int(p[3:5], 16))
                continue
            if pending:
                out.append(pending.decode('utf-8', errors='replace')); pending = bytearray()
            if i is not None and self.pieces[i][2] == CONTROL:
                continue
            if i is not None and self.pieces[i][2] == UNKNOWN:
                out.append(' ⁇ ')
                continue
            out.append(p.replace(SPACE, ' '))
        if pending:
            out.append(pending.decode('utf-8', errors='replace'))
        return ''.join(out)

    def decode(self, ids):
        return self.decode_pieces([self.pieces[i][0] for i in ids])


# ---- the reference's HF wrapper, restated ---------------------------------------------------------------------------
class InternLM2Tokenizer:
    """Drop-in for `AutoTokenizer.from_pretrained(INTERNVL_PATH, trust_remote_code=True)` on the chat path:
    `tok(query, return_tensors='pt')`, `convert_tokens_to_ids`, `batch_decode(..., skip_special_tokens=True)`."""

    def __init__(self, vocab_file, added_tokens=None, special_tokens=(), add_bos_token=True, add_eos_token=False):
        self.sp_model = SentencePieceBPE(vocab_file)
        self.add_bos_token, self.add_eos_token = add_bos_token, add_eos_token
        self.bos_token_id, self.eos_token_id = self.sp_model.bos_id, self.sp_model.eos_id
        self.added_tokens_encoder = dict(added_tokens or {})                 # content -> id
        self.added_tokens_decoder = {i: t for t, i in self.added_tokens_encoder.items()}
        self._split_tokens = sorted(self.added_tokens_encoder, key=len, reverse=True)
        self._split_re = re.compile('(' + '|'.join(re.escape(t) for t in self._split_tokens) + ')') if self._split_tokens else None
        self.all_special_ids = {self.sp_model.unk_id, self.bos_token_id, self.eos_token_id}
        self.all_special_ids.update(self.added_tokens_encoder[t] for t in special_tokens if t in self.added_tokens_encoder)
        self.padding_side = 'right'

    @classmethod
    def from_pretrained(cls, path, **kw):
        """Reads tokenizer.model + tokenizer_config.json / added_tokens.json / special_tokens_map.json of the checkpoint dir."""
        added, special = {}, set()
        cfg = os.path.join(path, 'tokenizer_config.json')
        if os.path.exists(cfg):
            c = json.load(open(cfg, encoding='utf-8'))
            for i, d in c.get('added_tokens_decoder', {}).items():
                added[d['content']] = int(i)
                if d.get('special'):
                    special.add(d['content'])
            special.update(t if isinstance(t, str) else t.get('content') for t in c.get('additional_special_tokens', []) or [])
        at = os.path.join(path, 'added_tokens.json')
        if os.path.exists(at):
            for t, i in json.load(open(at, encoding='utf-8')).items():
                added.setdefault(t, int(i))
        sm = os.path.join(path, 'special_tokens_map.json')
        if os.path.exists(sm):
            for v in json.load(open(sm, encoding='utf-8')).values():
                for t in (v if isinstance(v, list) else [v]):
                    special.add(t if isinstance(t, str) else t.get('content'))
        return cls(os.path.join(path, 'tokenizer.model'), added, special, **kw)

    @property
    def vocab_size(self):
        return len(self.sp_model)

    def _split_on_added(self, text):
        """[(chunk, id-or-None)]: longest added token at each position wins; the rest goes to sentencepiece.  One compiled alternation, longest first,
        split in C: a page prompt is 2 816 x <IMG_CONTEXT> in 40 KB of text, and a per-character Python scan of it cost 10 ms per page."""
        if self._split_re is None:
            return [(text, None)] if text else []
        enc = self.added_tokens_encoder
        parts = self._split_re.split(text)                         # [chunk, token, chunk, token, ..., chunk]
        return [(c, enc.get(c) if i & 1 else None) for i, c in enumerate(parts) if c or i & 1]

    def encode(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if (add_special_tokens and self.add_bos_token) else []
        for chunk, tid in self._split_on_added(text):
            if tid is not None:
                ids.append(tid)
            else:
                ids.extend(self.sp_model.encode(chunk))
        if add_special_tokens and self.add_eos_token:
            ids.append(self.eos_token_id)
        return ids

    def __call__(self, text, return_tensors=None, padding=False, add_special_tokens=True):
        batch = [text] if isinstance(text, str) else list(text)
        enc = [self.encode(t, add_special_tokens) for t in batch]
        if len(enc) > 1 and len({len(e) for e in enc}) > 1:
            if not padding:
                raise ValueError('sequences of different length need padding=True')
            m = max(len(e) for e in enc)
            pad = self.eos_token_id                                          # pad_token='</s>' (tokenization_internlm2.py:54)
            mask = [([0] * (m - len(e)) + [1] * len(e)) if self.padding_side == 'left' else ([1] * len(e) + [0] * (m - len(e))) for e in enc]
            enc = [([pad] * (m - len(e)) + e) if self.padding_side == 'left' else (e + [pad] * (m - len(e))) for e in enc]
        else:
            mask = [[1] * len(e) for e in enc]
        if return_tensors == 'pt':
            return {'input_ids': torch.tensor(enc, dtype=torch.long), 'attention_mask': torch.tensor(mask, dtype=torch.long)}
        if isinstance(text, str):
            return {'input_ids': enc[0], 'attention_mask': mask[0]}
        return {'input_ids': enc, 'attention_mask': mask}

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            if tokens in self.added_tokens_encoder:
                return self.added_tokens_encoder[tokens]
            return self.sp_model.piece_to_id.get(tokens, self.sp_model.unk_id)
        return [self.convert_tokens_to_ids(t) for t in tokens]

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self.added_tokens_decoder.get(ids) or self.sp_model.pieces[ids][0]
        return [self.convert_ids_to_tokens(int(i)) for i in ids]

    def decode(self, ids, skip_special_tokens=False):
        """transformers 4.45 PreTrainedTokenizer._decode (slow tokenizers): runs of ordinary ids go through the
        sentencepiece decoder, added tokens are kept verbatim, the pieces are joined with single spaces
        (spaces_between_special_tokens=True, clean_up_tokenization_spaces=False)."""
        ids = [int(i) for i in (ids.tolist() if hasattr(ids, 'tolist') else ids)]
        sub_texts, cur = [], []
        for i in ids:
            if skip_special_tokens and i in self.all_special_ids:
                continue
            if i in self.added_tokens_decoder:
                if cur:
                    sub_texts.append(self.sp_model.decode(cur)); cur = []
                sub_texts.append(self.added_tokens_decoder[i])
            else:
                cur.append(i)
        if cur:
            sub_texts.append(self.sp_model.decode(cur))
        return ' '.join(sub_texts)

    def batch_decode(self, sequences, skip_special_tokens=False):
        return [self.decode(s, skip_special_tokens) for s in sequences]
