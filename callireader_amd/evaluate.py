"""CalliBench `full_page` runner on the HIP engine: mirror of /root/reference/evaluate.py for the path's own task.

  reference                                             here
  evaluate.py:52-53   get_clean_string                   get_clean_string (same two character classes)
  evaluate.py:55-76   get_parquet                        get_parquet (columns `annotation` JSON, `image`.bytes)
  utils/utils.py:516-542 calculate_metrics               calculate_metrics (greedy one-to-one match = multiset intersection)
  evaluate.py:145-147 Levenshtein.distance / max len     edit_distance (third-party `Levenshtein` 0.x: plain unit-cost
                                                          edit distance on the two character lists; restated, the wheel is
                                                          not in this image)
  evaluate.py:127-132 single_rec -> cc.convert(response) single_rec; traditional -> simplified through `opencc` when it is
                                                          importable (reference: opencc.OpenCC('t2s.json')), identity
                                                          otherwise, and the JSON says which
  evaluate.py:134-171 test_full_page                      test_full_page (same arguments, same JSON layout) + `batch_pages`
  evaluate.py:389-436 main --type full_page               main (easy / medium / hard parquet files, prompt 读出图中所有文字。)

  (new) --fp8_decode / --fp8_mfma                          BASELINE config 5's fp8 switches (off by default)
  (new) --compare_fp8                                      the accuracy gate of those switches as ONE command: every file runs twice, bf16 then fp8,
                                                          and full_page_<level>_fp8_vs_bf16.json holds the per-page agreement and the F1 / NED deltas
  (new) one process per GPU under torchrun                 every rank takes a contiguous share of each file's pages, rows gathered on rank 0

  evaluate.py:173-213 test_region_wise                    test_region_wise (crop `region`, score against `answer`)
  evaluate.py:78-123  evaluate_accuracy                   evaluate_accuracy (pinned to the reference's function: tests/golden/eval_vectors.json)
  evaluate.py:216-313 test_choice                         test_choice (two turns: transcription, then the multiple-choice question with
                                                          the first turn as history; the reference scores the first 3 samples only, :257,301
                                                          -- `limit=3` by default here, `--choice_limit 0` lifts it)
  evaluate.py:317-386 test_bilingual / test_intent        the same two turns, answers stored for the reference's external judges (eval/: an STS
                                                          model and an LLM API, not part of this package)
`batch_pages` > 1 sends that many pages through the engine together, two batches in flight
(InternVLChatModel.chat_ocr_stream): each page's response equals its own chat_ocr call, pages/s is what changes.
"""
import argparse
import json
import os
import re
import sys
from collections import Counter
from io import BytesIO

from PIL import Image

Image.MAX_IMAGE_PIXELS = None

_ZH_PUNCT = re.compile('[。？！、，「」『』‘’“”–—…【】《》：；]')
_EN_PUNCT = re.compile(r'[,\.!?:\'";\(\)\[\]\{\}\-\n\*1234567890]')


def get_clean_string(text):
    """evaluate.py:42-53: ASCII punctuation, digits, newlines and `*` go first, then the CJK punctuation set."""
    return _ZH_PUNCT.sub('', _EN_PUNCT.sub('', text))


def get_parquet(parquet_path):
    """evaluate.py:55-76: (images, annotations); a row that fails to parse is reported and skipped."""
    import pandas as pd
    df = pd.read_parquet(parquet_path)
    images, annotations = [], []
    for index, row in df.iterrows():
        try:
            labels = json.loads(row['annotation'])
            image = Image.open(BytesIO(row['image']['bytes']))
            images.append(image)
            annotations.append(labels)
        except Exception as e:
            print(f'Row {index} Error: {e}')
            continue
    return images, annotations


def calculate_metrics(y_pred, y_gt):
    """utils/utils.py:516-542 with the default equality comparison: every predicted item claims the first unclaimed equal
    ground-truth item, so TP is the size of the multiset intersection; FP = |pred| - TP, FN = |gt| - TP."""
    cp, cg = Counter(y_pred), Counter(y_gt)
    tp = sum(min(n, cg[k]) for k, n in cp.items())
    fp, fn = len(y_pred) - tp, len(y_gt) - tp
    precision = tp / (tp + fp) if tp + fp > 0 else 0
    recall = tp / (tp + fn) if tp + fn > 0 else 0
    f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) > 0 else 0
    return precision, recall, f1


def edit_distance(a, b):
    """Unit-cost insert / delete / substitute distance between two sequences (what Levenshtein.distance returns)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


class _T2S:
    """opencc.OpenCC('t2s.json') when the package is there; the reference cannot run without it, this runner can."""

    def __init__(self):
        try:
            import opencc
            self._cc, self.name = opencc.OpenCC('t2s.json'), 'opencc t2s'
        except Exception:
            self._cc, self.name = None, 'identity (opencc not installed)'
            print('[callireader_amd.evaluate] WARNING: opencc is not installed -- responses are NOT converted traditional -> simplified, so P / R / '
                  'F1 / NED are not comparable with the reference\'s evaluate.py numbers', file=sys.stderr)

    def convert(self, text):
        return self._cc.convert(text) if self._cc is not None else text


cc = _T2S()


def single_rec(model, tokenizer, detect_model, generation_config, image_path, prompt, use_p, hard_vq, drop_zero, repetition_penalty, verbose):
    response, history = model.chat_ocr(tokenizer, detect_model, image_path, prompt, generation_config, use_p=use_p, hard_vq=hard_vq,
                                       drop_zero=drop_zero, repetition_penalty=repetition_penalty, return_history=True, verbose=verbose)
    return cc.convert(response)


def score_page(response, reference):
    """evaluate.py:141-149 for one page: character lists, P / R / F1 and the normalised edit distance."""
    gt = list(get_clean_string(reference))
    response = list(response)
    precision, recall, f1 = calculate_metrics(response, gt)
    max_len = max(len(response), len(gt))
    ned = edit_distance(response, gt) / max_len if max_len else 0.0
    return response, gt, precision, recall, f1, ned


def test_full_page(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, prompt, use_p, hard_vq, drop_zero,
                   repetition_penalty, verbose, batch_pages=1, limit=None, shard=None, gather=None):
    """shard = (rank, world): this process takes a contiguous share of the pages (one process per GPU, BASELINE config 5);
    gather(list) -> list of all ranks' lists in rank order (e.g. torch.distributed.all_gather_object); the report holds every page
    in file order and is written by rank 0."""
    images, annotations = get_parquet(parquet_path)
    if limit is not None:
        images, annotations = images[:limit], annotations[:limit]
    rank, world = shard if shard is not None else (0, 1)
    if world > 1:
        lo, hi = len(images) * rank // world, len(images) * (rank + 1) // world
        images, annotations = images[lo:hi], annotations[lo:hi]
    to_be_save = {'detailed': []}
    sums = [0.0, 0.0, 0.0, 0.0]
    count = 0
    step = max(batch_pages, 1)
    groups = [(images[i0:i0 + step], annotations[i0:i0 + step]) for i0 in range(0, len(images), step)]
    if batch_pages > 1:
        # two batches in flight: batch i decodes while batch i+1 goes through detection, tiling, the visual stage and the prefill
        answers = model.chat_ocr_stream(tokenizer, detect_model, (g[0] for g in groups), prompt, generation_config, use_p=use_p, hard_vq=hard_vq,
                                        drop_zero=drop_zero, repetition_penalty=repetition_penalty)
    else:
        answers = ([single_rec(model, tokenizer, detect_model, generation_config, g[0][0], prompt, use_p, hard_vq, drop_zero, repetition_penalty, verbose)]
                   for g in groups)
    for (imgs, annots), responses in zip(groups, answers):
        responses = [cc.convert(r) for r in responses] if batch_pages > 1 else responses
        for response, annot in zip(responses, annots):
            response, gt, precision, recall, f1, ned = score_page(response, annot['reference'])
            to_be_save['detailed'].append({'imgPath': annot['imagePath'], 'prompt': prompt, 'output': ''.join(response), 'gt': ''.join(gt),
                                           'precision': precision, 'recall': recall, 'f1': f1, 'ned': ned})
            for k, v in enumerate((precision, recall, f1, ned)):
                sums[k] += v
            count += 1
    if world > 1:
        parts = gather(to_be_save['detailed'])
        to_be_save['detailed'] = [row for part in parts for row in part]
        count = len(to_be_save['detailed'])
        sums = [sum(row[k] for row in to_be_save['detailed']) for k in ('precision', 'recall', 'f1', 'ned')]
    avg = [s / count for s in sums] if count else sums
    to_be_save['average'] = {'ave_precison': avg[0], 'avg_recall': avg[1], 'avg_f1': avg[2], 'avg_ned': avg[3]}     # keys as upstream (sic)
    to_be_save['t2s'] = cc.name
    if rank == 0:
        with open(save_json_path, 'w', encoding='utf-8') as f:
            json.dump(to_be_save, f, ensure_ascii=False, indent=4)
    return to_be_save['average']


def _crop_region(img, region):
    import numpy as np
    (x1, y1), (x2, y2) = region
    return Image.fromarray(np.array(img.convert('RGB'))[y1:y2, x1:x2])


def test_region_wise(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, prompt, use_p, hard_vq, drop_zero,
                     repetition_penalty, verbose, limit=None):
    """evaluate.py:173-213: every row names a `region` [[x1, y1], [x2, y2]] of its page; the crop goes through chat_ocr like a page and
    is scored against `answer` with the full-page metrics."""
    images, annotations = get_parquet(parquet_path)
    if limit:
        images, annotations = images[:limit], annotations[:limit]
    rows, sums = [], [0.0, 0.0, 0.0, 0.0]
    for img, annot in zip(images, annotations):
        response = single_rec(model, tokenizer, detect_model, generation_config, _crop_region(img, annot['region']), prompt, use_p, hard_vq, drop_zero,
                              repetition_penalty, verbose)
        response, gt, precision, recall, f1, ned = score_page(response, annot['answer'])
        rows.append({'imgPath': annot['imagePath'], 'prompt': prompt, 'output': ''.join(response), 'gt': ''.join(gt),
                     'precision': precision, 'recall': recall, 'f1': f1, 'ned': ned})
        for k, v in enumerate((precision, recall, f1, ned)):
            sums[k] += v
    n = max(len(rows), 1)
    report = {'detailed': rows, 'average': {'ave_precison': sums[0] / n, 'avg_recall': sums[1] / n, 'avg_f1': sums[2] / n, 'avg_ned': sums[3] / n}, 't2s': cc.name}
    with open(save_json_path, 'w', encoding='utf-8') as f:
        json.dump(report, f, ensure_ascii=False, indent=4)
    return report['average']


def evaluate_accuracy(responses, correct_answers):
    """evaluate.py:78-123.  correct_answers[i] = (letter, its text, the two other options' texts).  A response counts when it names
    exactly the right letter among A / B / C -- unless it quotes an option's TEXT: quoting the right text and neither wrong one
    counts whatever the letters say, quoting the right text and a wrong one does not."""
    assert len(responses) == len(correct_answers), 'Responses and answers must have the same length.'
    right = 0
    for response, (letter, text_gt, wrong0, wrong1) in zip(responses, correct_answers):
        named = [c for c in 'ABC' if c in response]
        ok = len(named) <= 1 and (named[0] if named else None) == letter
        if text_gt in response:
            ok = not (wrong0 in response or wrong1 in response)
        right += bool(ok)
    return right / len(responses) * 100


def parse_choice(prompt, letter):
    """evaluate.py:235-257: the option lines of a multiple-choice prompt ("A: ...", "B: ...", "C: ..."): returns
    (letter, the right option's text, the two other options' texts in prompt order)."""
    right, wrong = None, []
    for line in prompt.split('\n'):
        if not any(c in line for c in 'ABC'):
            continue
        if line.startswith(letter + ':'):
            right = line
        elif len(wrong) < 2:
            wrong.append(line)
    return (letter, right.split(':')[1].strip(), wrong[0].split(':')[1].strip(), wrong[1].split(':')[1].strip())


FIRST_TURN = '这幅书法作品内容是什么？'


def _two_turns(model, tokenizer, detect_model, generation_config, img, second_question, use_p, hard_vq, drop_zero, repetition_penalty, verbose):
    """The reference's reasoning tasks ask for the transcription first and put the actual question as a second turn on that history."""
    kw = dict(use_p=use_p, hard_vq=hard_vq, drop_zero=drop_zero, repetition_penalty=repetition_penalty, return_history=True, verbose=verbose)
    _, history = model.chat_ocr(tokenizer, detect_model, img, FIRST_TURN, generation_config, **kw)
    response, history = model.chat_ocr(tokenizer, detect_model, img, second_question, generation_config, history=history, **kw)
    return response


def test_choice(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, use_p=True, hard_vq=False, drop_zero=True,
                repetition_penalty=1.0, verbose=False, limit=3):
    """evaluate.py:216-313 (author / style / layout).  The reference scores the first three samples of each file (:257, :301); limit=0
    scores all."""
    images, annotations = get_parquet(parquet_path)
    answers = [parse_choice(a['conversations'][0]['value'], a['conversations'][1]['value']) for a in annotations]
    if limit:
        images, annotations, answers = images[:limit], annotations[:limit], answers[:limit]
    responses, rows = [], []
    for img, annot, gt in zip(images, annotations, answers):
        prompt = annot['conversations'][0]['value'].replace('<image>\n', '')
        response = _two_turns(model, tokenizer, detect_model, generation_config, img, prompt + '\n只需要输出问题的答案，禁止输出其他内容！答案：',
                              use_p, hard_vq, drop_zero, repetition_penalty, verbose)
        responses.append(response)
        rows.append({'imgPath': annot['image'], 'output': response, 'reference': gt[0]})
    accuracy = evaluate_accuracy(responses, answers) if responses else 0.0
    report = {'detailed': rows, 'summary': {'total_samples': len(responses), 'accuracy': accuracy}}
    with open(save_json_path, 'w', encoding='utf-8') as f:
        json.dump(report, f, ensure_ascii=False, indent=4)
    return accuracy, report


def _test_reasoning(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, use_p, hard_vq, drop_zero, repetition_penalty,
                    verbose, with_content, limit=None):
    """evaluate.py:317-386: question = the prompt up to its first option line; the answer is split at "INTENT:" and stored for the
    external judges of eval/ (not run here)."""
    images, annotations = get_parquet(parquet_path)
    if limit:
        images, annotations = images[:limit], annotations[:limit]
    rows = []
    for img, annot in zip(images, annotations):
        prompt = annot['conversations'][0]['value']
        m = re.search(r'^(.*?)\n[A-Z]:', prompt, re.DOTALL)
        question = m.group(1).strip() if m else prompt
        response = _two_turns(model, tokenizer, detect_model, generation_config, img, question, use_p, hard_vq, drop_zero, repetition_penalty, verbose)
        row = {'imgPath': annot['image'], 'chinese': response.split('INTENT:')[0], 'answer': response.split('INTENT:')[-1],
               'gt': annot['conversations'][-1]['value']}
        if with_content:
            row['calligraphy_content'] = annot['content']
        rows.append(row)
    with open(save_json_path, 'w', encoding='utf-8') as f:
        json.dump({'detailed': rows}, f, ensure_ascii=False, indent=4)
    return rows


def test_bilingual(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, use_p, hard_vq, drop_zero, repetition_penalty, verbose, limit=None):
    return _test_reasoning(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, use_p, hard_vq, drop_zero, repetition_penalty, verbose, False, limit)


def test_intent(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, use_p, hard_vq, drop_zero, repetition_penalty, verbose, limit=None):
    return _test_reasoning(parquet_path, save_json_path, model, tokenizer, detect_model, generation_config, use_p, hard_vq, drop_zero, repetition_penalty, verbose, True, limit)


def compare_reports(bf16_rows, fp8_rows):
    """--compare_fp8: the same pages through the bf16 path and through the fp8 switches -> what changed.  Rows are the `detailed`
    entries test_full_page writes (same order).  The gate BASELINE config 5 needs is delta_f1 / delta_ned on real weights."""
    assert len(bf16_rows) == len(fp8_rows)
    n = len(bf16_rows)
    same, ned_between = 0, 0.0
    for a, b in zip(bf16_rows, fp8_rows):
        assert a['imgPath'] == b['imgPath']
        same += a['output'] == b['output']
        m = max(len(a['output']), len(b['output']))
        ned_between += edit_distance(list(a['output']), list(b['output'])) / m if m else 0.0

    def avg(rows, k):
        return sum(r[k] for r in rows) / n if n else 0.0
    out = {'pages': n, 'identical_outputs': same, 'mean_ned_between_outputs': ned_between / n if n else 0.0}
    for k in ('precision', 'recall', 'f1', 'ned'):
        out[f'avg_{k}_bf16'], out[f'avg_{k}_fp8'] = avg(bf16_rows, k), avg(fp8_rows, k)
        out[f'delta_{k}'] = out[f'avg_{k}_fp8'] - out[f'avg_{k}_bf16']
    return out


def main(argv=None):
    parser = argparse.ArgumentParser(description='args for inference task')
    parser.add_argument('--type', type=str, choices=['full_page', 'region_wise', 'choice', 'bilingual', 'intent'], default='full_page',
                        help='Evaluation Type (full_page, region_wise, choice, bilingual, intent)')
    parser.add_argument('--choice_limit', type=int, default=3, help='samples scored per choice file (the reference scores 3; 0 = all)')
    parser.add_argument('--save_name', type=str, default='exp')
    parser.add_argument('--data', type=str, default='./CalliBench', help='Evaluation Data Directory')
    parser.add_argument('--use_p', type=bool, default=True)
    parser.add_argument('--hard_vq', type=bool, default=False)
    parser.add_argument('--drop_zero', type=bool, default=False)
    parser.add_argument('--verbose', type=bool, default=False)
    parser.add_argument('--repetition_penalty', type=float, default=1.0)
    parser.add_argument('--model', type=str, default='InternVL', help='checkpoint dir (INTERNVL_PATH)')
    parser.add_argument('--params', type=str, default='./params')
    parser.add_argument('--batch_pages', type=int, default=64, help='pages sent through the engine together (64 = the most rows the batched decode takes, the batch the headline is measured on; 1 = page after page, as the reference)')
    parser.add_argument('--fp8_decode', action='store_true', help='BASELINE config 5: e4m3 weights for the batched decode (cr_enable_fp8_decode)')
    parser.add_argument('--fp8_mfma', type=int, nargs='?', const=1, default=0, choices=(0, 1, 2),
                        help='BASELINE config 5: e4m3 x e4m3 matrix-core linears (cr_enable_fp8_mfma): 1 = the norm-fed linears of the ViT / projector / prefill, 2 = also ViT fc2 and the prefill\'s wo / w2')
    parser.add_argument('--compare_fp8', action='store_true', help='run every file twice -- bf16, then with --fp8_decode / --fp8_mfma (both when neither is given) -- and '
                                                                   'write full_page_<level>_fp8_vs_bf16.json: identical outputs, NED between them, F1 / NED deltas')
    parser.add_argument('--allow_no_t2s', action='store_true', help='run without opencc (responses stay unconverted: scores not comparable with the reference\'s)')
    args = parser.parse_args(argv)
    if cc._cc is None and not args.allow_no_t2s:
        raise SystemExit('opencc is not installed: the reference converts every response traditional -> simplified before scoring (evaluate.py:127-132). '
                         'Install opencc, or pass --allow_no_t2s to score unconverted text (not comparable with the reference\'s numbers).')
    import torch
    from .inference import load_detector
    from .modeling_internvl_chat import InternVLChatModel
    from .tokenization_internlm2 import InternLM2Tokenizer
    save_dir = f'outputs/{args.save_name}'
    os.makedirs(save_dir, exist_ok=True)
    # one process per GPU (torchrun --nproc-per-node N -m callireader_amd.evaluate ...): every rank takes its share of each file's pages
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    gather = None
    if world > 1 and args.type != 'full_page':
        # these tasks are page-at-a-time host loops (two turns, up to 1024 tokens each) that only rank 0 would run while the others sat in a
        # barrier until the collective timeout aborted the job: refuse before any rank loads a model or joins a process group
        raise SystemExit(f'--type {args.type} runs in one process (it is a serial loop over chat_ocr, evaluate.py:438-466); start it without torchrun. '
                         'Only --type full_page shards its pages over ranks.')
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get('CR_DIST_BACKEND', 'nccl'), device_id=torch.device('cuda', local))

        def gather(rows):
            out = [None] * world
            dist.all_gather_object(out, rows)
            return out
    model = InternVLChatModel.from_pretrained(args.model, params_dir=args.params, torch_dtype=torch.bfloat16, max_pages=args.batch_pages, device=local).eval().cuda()
    want_mfma, want_dec = args.fp8_mfma, args.fp8_decode
    if args.compare_fp8 and not (want_mfma or want_dec):
        want_mfma, want_dec = 1, True

    def fp8(on):
        if want_mfma:
            model.engine.enable_fp8_mfma(on, level=want_mfma)
        if want_dec:
            model.engine.enable_fp8_decode(on)
    if not args.compare_fp8:
        fp8(True)
    tokenizer = InternLM2Tokenizer.from_pretrained(args.model)
    generation_config = dict(num_beams=1, max_new_tokens=1024, do_sample=False)
    detect_model = load_detector(args.params)
    common = (args.use_p, args.hard_vq, args.drop_zero, args.repetition_penalty, args.verbose)
    if args.type != 'full_page':
        # the other CalliBench tasks are page-at-a-time host loops over chat_ocr (one process; evaluate.py:438-466)
        if True:
            if args.type == 'region_wise':
                print(test_region_wise(os.path.join(args.data, 'region-wise/region.parquet'), os.path.join(save_dir, 'region_wise.json'), model, tokenizer,
                                       detect_model, generation_config, '读出图中区域所有文字。', *common))
            elif args.type == 'choice':
                for name in ('author', 'style', 'layout'):
                    acc, _ = test_choice(os.path.join(args.data, f'choice/{name}/{name}.parquet'), os.path.join(save_dir, f'{name}.json'), model, tokenizer,
                                         detect_model, generation_config, *common, limit=args.choice_limit)
                    print(name, acc)
            elif args.type == 'bilingual':
                test_bilingual(os.path.join(args.data, 'reasoning/bilingual/medium/bilingual_medium.parquet'), os.path.join(save_dir, 'bilingual.json'), model,
                               tokenizer, detect_model, generation_config, *common)
            else:
                test_intent(os.path.join(args.data, 'reasoning/intent/intent.parquet'), os.path.join(save_dir, 'intent.json'), model, tokenizer, detect_model,
                            generation_config, *common)
        return
    prompt = '读出图中所有文字。'
    for level in ('easy', 'medium', 'hard'):
        parquet_path = os.path.join(args.data, f'full_page_ocr/{level}/{level}.parquet')
        save_json_path = os.path.join(save_dir, f'full_page_{level}.json')
        run = lambda path: test_full_page(parquet_path, path, model, tokenizer, detect_model, generation_config, prompt, args.use_p, args.hard_vq,   # noqa: E731
                                          args.drop_zero, args.repetition_penalty, args.verbose, batch_pages=args.batch_pages, shard=(rank, world), gather=gather)
        avg = run(save_json_path)
        if rank == 0:
            print(level, avg)
        if args.compare_fp8:
            fp8_path = os.path.join(save_dir, f'full_page_{level}_fp8.json')
            fp8(True)
            try:
                avg8 = run(fp8_path)
            finally:
                fp8(False)
            if rank == 0:
                cmp = compare_reports(json.load(open(save_json_path, encoding='utf-8'))['detailed'], json.load(open(fp8_path, encoding='utf-8'))['detailed'])
                cmp['switches'] = {'fp8_mfma': want_mfma, 'fp8_decode': want_dec}
                with open(os.path.join(save_dir, f'full_page_{level}_fp8_vs_bf16.json'), 'w', encoding='utf-8') as f:
                    json.dump(cmp, f, ensure_ascii=False, indent=4)
                print(level, 'fp8', avg8, 'identical outputs', cmp['identical_outputs'], 'of', cmp['pages'], 'delta F1', cmp['delta_f1'], 'delta NED', cmp['delta_ned'])
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
