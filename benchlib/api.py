"""`api_level`: the headline's batch through the reference's own entry points, from image FILES to decoded strings.

bench.py's headline step starts from tiles resident in HBM (the contract's `value`).  What a user of the reference calls is
`inference.py --tgt <folder>` -> `folder_rec` -> `model.chat_ocr(...)` per image (/root/reference/inference.py:47-62).  This block runs, at full
InternVL2-8B dims on the same model object:
  * `model.chat_ocr_stream(tokenizer, None, batches of 64 file paths, prompt, gen, boxes_batches=...)`: JPEG decode, tiling on the GPU from one upload per page,
    the real tokenizer class on a checkpoint-dir layout, visual stage, splice, prefill, batched decode, ids -> strings;
  * `callireader_amd.inference.folder_rec(...)` on a folder of 128 such files (boxes from the JSON next to each image: there is no detector network in this image).
Pages are tests/golden/example0.jpg (the reference's examples/0.jpg: 788x2000, 11 page tiles) with its 96 boxes; the tokenizer is a sentencepiece BPE model trained
here in a second (the reference's tokenizer.model does not travel) with the reference's added-token ids, so the prompt has the reference's structure and ~its length."""
import json
import os
import random
import shutil
import tempfile
import time

import torch

VOCAB_SP = 92544                                                       # pieces in the reference's tokenizer.model; [UNUSED_TOKEN_k] = 92397 + k (user-defined pieces)
ADDED = {'<|plugin|>': 92538, '<|interpreter|>': 92539, '<|action_end|>': 92540, '<|action_start|>': 92541, '<|im_end|>': 92542, '<|im_start|>': 92543, '<img>': 92544,
         '</img>': 92545, '<IMG_CONTEXT>': 92546, '<quad>': 92547, '</quad>': 92548, '<ref>': 92549, '</ref>': 92550, '<box>': 92551, '</box>': 92552}      # InternVL/tokenizer_config.json: added_tokens_decoder
PROMPT = '这幅书法作品内容是什么？'                                     # inference.py:69


def make_tokenizer_dir(d):
    """tokenizer.model + tokenizer_config.json in the reference's checkpoint-dir layout: a BPE model trained here, filled up to the reference's 92 544 pieces
    (random-init weights pick any id of the vocabulary: every id must decode) with its user-defined [UNUSED_TOKEN_k] pieces where the reference has them
    ([UNUSED_TOKEN_140] = 92 537, the pseudo-token placeholder, modeling_internvl_chat.py:1100), and the reference's added-token ids."""
    import sentencepiece as spm
    from sentencepiece import sentencepiece_model_pb2 as pb
    rng = random.Random(0)
    alphabet = ['abcdefghijklmnopqrstuvwxyz', ' ', ' ', '，。？！：', '这幅书法作品内容是什么读出图中所有文字你是由上海人工智能实验室联合商汤科技开发的书生多模态大模型英文名叫一个有用无害助手', 'ABCDEFGHIJInternVL', '\n']
    corpus = os.path.join(d, 'corpus.txt')
    with open(corpus, 'w', encoding='utf-8') as f:
        for _ in range(3000):
            f.write(''.join(rng.choice(rng.choice(alphabet)) for _ in range(rng.randint(5, 60))).replace('\n', ' ') + '\n')
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(d, 'tokenizer'), vocab_size=600, model_type='bpe', character_coverage=0.995,
                                   normalization_rule_name='identity', add_dummy_prefix=False, remove_extra_whitespaces=False, byte_fallback=True, minloglevel=2)
    m = pb.ModelProto()
    m.ParseFromString(open(os.path.join(d, 'tokenizer.model'), 'rb').read())
    first_unused = VOCAB_SP - 147
    for i in range(len(m.pieces), VOCAB_SP):
        m.pieces.add(piece=f'[UNUSED_TOKEN_{i - first_unused}]' if i >= first_unused else f'\u2581filler{i}', score=0.0 if i >= first_unused else -1e4 - i, type=4 if i >= first_unused else 1)
    open(os.path.join(d, 'tokenizer.model'), 'wb').write(m.SerializeToString())
    dec = {'0': {'content': '<unk>', 'special': True}, '1': {'content': '<s>', 'special': True}, '2': {'content': '</s>', 'special': True}}
    for t, i in ADDED.items():
        dec[str(i)] = {'content': t, 'special': True}
    json.dump({'added_tokens_decoder': dec}, open(os.path.join(d, 'tokenizer_config.json'), 'w'))
    os.remove(corpus)


def make_pages(d, n, root):
    """n copies of the example page + the labelme-style boxes JSON next to each (what inference.py reads when there is no detector); -> (paths, boxes)"""
    from callireader_amd.preprocess import boxes_from_labelme
    src, bj = os.path.join(root, 'tests', 'golden', 'example0.jpg'), os.path.join(root, 'tests', 'golden', 'example0_boxes.json')
    boxes = boxes_from_labelme(json.load(open(bj, encoding='utf-8')))
    paths = []
    for k in range(n):
        p = os.path.join(d, f'page_{k:04d}.jpg')
        shutil.copyfile(src, p)
        shutil.copyfile(bj, os.path.join(d, f'page_{k:04d}.json'))
        paths.append(p)
    return paths, boxes


class _Box:
    def __init__(self, b):
        self.xyxy = torch.tensor([b], dtype=torch.float32)


class _Result:
    def __init__(self, boxes):
        self.boxes = [_Box(b) for b in boxes]


class _DetectorStandIn:
    """ultralytics.YOLO's call shape; the boxes of the example page in an order that is not the reading order."""

    def __init__(self, boxes):
        self.boxes = [list(b) for b in boxes]
        random.Random(1).shuffle(self.boxes)
        self.result = [_Result(self.boxes)]

    def __call__(self, image, verbose=True):
        return self.result


def _with_detector(model, tok, batch_paths, boxes, gen, pages):
    from callireader_amd import synthetic
    had = model.sorter
    if had is None:
        model.load_orderformer(synthetic.make_orderformer_state_dict(seed=11))
    det = _DetectorStandIn(boxes)
    try:
        list(model.chat_ocr_stream(tok, det, batch_paths[:1], PROMPT, gen, repetition_penalty=1.0))          # warm-up (the sorter's first launches)
        stats = {}
        t0 = time.perf_counter()
        ts, n = [], 0
        for res in model.chat_ocr_stream(tok, det, batch_paths, PROMPT, gen, repetition_penalty=1.0, stats=stats):
            ts.append(time.perf_counter())
            n += len(res)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        if had is None:
            model.sorter = None
    gaps = sorted(b - a for a, b in zip(ts, ts[1:]))
    host = stats.get('host', {})
    return {'what': 'chat_ocr_stream(tokenizer, detect_model, batches of file paths, prompt, gen): a detector OBJECT with the ultralytics call shape (a stand-in that returns the example page\'s 96 boxes '
                    'shuffled; the YOLO network is third-party) and everything behind it per page: de-duplication, columns, OrderFormer on the GPU (seeded weights), reading order',
            'pages': n, 'batches': len(batch_paths), 'wall_s': round(wall, 3), 'pages_per_s': round(n / wall, 4),
            'steady_ms_per_batch': round(gaps[len(gaps) // 2] * 1e3, 1) if gaps else None,
            'ordering_front_end_ms_per_page': round(1e3 * host.get('detect_s', 0.0) / max(host.get('pages', 1), 1), 3),
            'compute_stream_idle_ms_between_batches': stats.get('compute_stream_idle_ms_between_batches')}


def api_level(model, root, pages=64, batches=4, new_tokens=128, folder_pages=128, headline_ms_per_step=None, headline_pages=None):
    from callireader_amd import inference as inf
    from callireader_amd.tokenization_internlm2 import InternLM2Tokenizer
    work = tempfile.mkdtemp(prefix='cr_api_', dir='/tmp')
    try:
        tokd = os.path.join(work, 'InternVL')
        os.makedirs(tokd)
        make_tokenizer_dir(tokd)
        tok = InternLM2Tokenizer.from_pretrained(tokd)
        stream_dir, folder_dir = os.path.join(work, 'stream'), os.path.join(work, 'folder')
        os.makedirs(stream_dir), os.makedirs(folder_dir)
        paths, boxes = make_pages(stream_dir, pages * batches, root)
        gen = dict(num_beams=1, max_new_tokens=new_tokens, do_sample=False)
        batch_paths = [paths[i:i + pages] for i in range(0, len(paths), pages)]
        boxes_batches = [[boxes] * len(b) for b in batch_paths]

        def run(bp, bb, stats=None):
            t0 = time.perf_counter()
            ts, out = [], []
            for res in model.chat_ocr_stream(tok, None, bp, PROMPT, gen, boxes_batches=bb, repetition_penalty=1.0, stats=stats):
                ts.append(time.perf_counter())
                out.append(res)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, ts, out
        run(batch_paths[:1], boxes_batches[:1])                          # warm-up: the feeder's context and threads, pinned buffers, the pipeline's caches
        stats = {}
        wall, ts, out = run(batch_paths, boxes_batches, stats)
        assert all(isinstance(r, str) for b in out for r in b) and sum(len(b) for b in out) == len(paths)
        n = len(paths)
        gaps = sorted(b - a for a, b in zip(ts, ts[1:]))
        steady = gaps[len(gaps) // 2] if gaps else None
        host = stats.get('host', {})
        per_page = {k[:-2] + '_ms_per_page': round(1e3 * v / max(host.get('pages', 1), 1), 3) for k, v in host.items() if k.endswith('_s')}
        ids = tok(model._build_query('<image>\n' + PROMPT + '[UNUSED_TOKEN_140]' * (3 * len(boxes)), None, [11], '<img>', '</img>', '<IMG_CONTEXT>')[0], return_tensors='pt')['input_ids']
        res = {'what': 'the same model object through the reference\'s API, from JPEG FILES to decoded strings: model.chat_ocr_stream(tokenizer, None, batches of file paths, '
                       'prompt, generation_config, boxes_batches=..., repetition_penalty=1.0) -- decode of the image files on threads one batch ahead, one upload per page, tiles cut on the GPU, '
                       'the engine\'s tokenizer class, visual stage, splice, prefill, batched greedy decode, ids -> text; pages = tests/golden/example0.jpg (788x2000, 11 page + 96 character tiles)',
               'pages': n, 'batches': len(batch_paths), 'pages_per_batch': pages, 'prompt_tokens': int(ids.numel()), 'new_tokens': new_tokens,
               'wall_s': round(wall, 3), 'pages_per_s': round(n / wall, 4),
               'steady_ms_per_batch': round(steady * 1e3, 1) if steady else None, 'steady_pages_per_s': round(pages / steady, 4) if steady else None,
               'host_ms_per_page': per_page,
               'host_note': 'decode runs on the feeder\'s threads (summed over threads, not on the critical path); decode_wait = what the feeding thread waited for them; '
                            'tokenize = prompt ids (skeleton + run expansion); plan = job tables; enqueue = uploads + tile kernels issued on the feeder\'s stream',
               'compute_stream_idle_ms_between_batches': stats.get('compute_stream_idle_ms_between_batches'),
               'idle_note': 'GPU time between the last kernel of batch i\'s prefill and the first kernel of batch i+1\'s visual stage on the compute stream (events)'}
        if headline_ms_per_step and headline_pages:
            res['vs_synthetic_headline'] = round((n / wall) / (headline_pages / (headline_ms_per_step * 1e-3)), 4)
            if steady:
                res['steady_vs_synthetic_headline'] = round((pages / steady) / (headline_pages / (headline_ms_per_step * 1e-3)), 4)
        # ---- one page through the reference's own call (config 3's shape, from the file): model.chat_ocr(tokenizer, None, path, prompt, gen, boxes=...) ----
        model.chat_ocr(tok, None, paths[0], PROMPT, gen, boxes=boxes, repetition_penalty=1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(3):
            model.chat_ocr(tok, None, paths[1 + k], PROMPT, gen, boxes=boxes, repetition_penalty=1.0)
        torch.cuda.synchronize()
        res['chat_ocr_single_page_s'] = round((time.perf_counter() - t0) / 3, 4)
        # ---- the reference's call shape WITH a detector object: detect_model(image array, verbose=False)[0].boxes[i].xyxy (inference.py:98, modeling_internvl_chat.py:356-362) ----
        # The YOLO network itself is third-party and not in this image: the stand-in returns the example's 96 boxes in a shuffled order, and everything behind the detector
        # runs as for real pages -- duplicate removal, column merge / 2-means split, the OrderFormer on the GPU (seeded weights), per-column assembly (ordering.py) -- on the
        # feeder's stream and context, page by page, before the tiles are cut.
        try:
            res['with_detector_object'] = _with_detector(model, tok, batch_paths[:min(2, len(batch_paths))], boxes, gen, pages)
        except Exception as e:
            res['with_detector_object'] = {'error': f'{type(e).__name__}: {e}'}
        # ---- the reference's folder mode (inference.py:47-62) on the batched path ----
        fpaths, _ = make_pages(folder_dir, folder_pages, root)
        save = os.path.join(work, 'recognition.json')
        import contextlib
        import io
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):         # folder_rec prints every response, as the reference does: stdout carries the bench line only
            results = inf.folder_rec(model, tok, None, gen, folder_dir, PROMPT, save, True, False, False, 1.0, False, batch_pages=pages)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert len(results) == folder_pages and all(r['response'] != 'ERROR!' for r in results) and os.path.exists(save)
        res['folder_rec'] = {'what': f'callireader_amd.inference.folder_rec (the body of `python inference.py --tgt <folder>`, model already loaded) on a folder of {folder_pages} such images, '
                                     f'boxes from the JSON next to each image, batch_pages={pages}: listing, {folder_pages // pages} batches through chat_ocr_stream (pipeline fill and drain included), results JSON written',
                             'pages': folder_pages, 'wall_s': round(dt, 3), 'pages_per_s': round(folder_pages / dt, 4)}
        if headline_ms_per_step and headline_pages:
            res['folder_rec']['vs_synthetic_headline'] = round((folder_pages / dt) / (headline_pages / (headline_ms_per_step * 1e-3)), 4)
        return res
    finally:
        shutil.rmtree(work, ignore_errors=True)
