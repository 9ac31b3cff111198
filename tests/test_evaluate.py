"""CalliBench full_page runner (callireader_amd/evaluate.py, mirror of the reference's evaluate.py:134-171,389-436):
scoring pinned to the reference's own functions (tests/golden/eval_vectors.json), the parquet reader and the JSON
report exercised on a synthetic two-page parquet with a stand-in model (CPU only)."""
import io
import json
import os

import pytest
from PIL import Image

from callireader_amd import evaluate as ev

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'eval_vectors.json'), encoding='utf-8'))


def test_metrics_equal_the_reference():
    for c in GOLD['metrics']:
        p, r, f = ev.calculate_metrics(list(c['pred']), list(c['gt']))
        assert (p, r, f) == (c['precision'], c['recall'], c['f1']), c
        assert ev.edit_distance(list(c['pred']), list(c['gt'])) == c['edit_distance_textbook'], c


def test_clean_string_equals_the_reference():
    for c in GOLD['clean']:
        assert ev.get_clean_string(c['in']) == c['out'], c


class FakeModel:
    def __init__(self, answers):
        self.answers, self.calls = list(answers), []

    def chat_ocr(self, tokenizer, detect_model, image, prompt, generation_config, **kw):
        self.calls.append((image.size, prompt, kw))
        return self.answers[len(self.calls) - 1], []

    def chat_ocr_stream(self, tokenizer, detect_model, image_batches, prompt, generation_config, **kw):
        for images in image_batches:                       # the engine's generator yields one list of responses per batch, in order
            self.calls.append((tuple(im.size for im in images), prompt, kw))
            n0 = sum(len(c[0]) if isinstance(c[0][0], tuple) else 1 for c in self.calls[:-1])
            yield self.answers[n0:n0 + len(images)]


def make_parquet(path, refs):
    pd = pytest.importorskip('pandas')
    rows = []
    for i, ref in enumerate(refs):
        buf = io.BytesIO()
        Image.new('RGB', (40 + i, 60), (255, 255, 255)).save(buf, format='PNG')
        rows.append({'annotation': json.dumps({'imagePath': f'p{i}.jpg', 'reference': ref}, ensure_ascii=False), 'image': {'bytes': buf.getvalue()}})
    rows.append({'annotation': '{broken json', 'image': {'bytes': b''}})           # a bad row is reported and skipped
    pd.DataFrame(rows).to_parquet(path)


@pytest.mark.parametrize('batch_pages', [1, 2])
def test_full_page_report(tmp_path, batch_pages):
    pq = str(tmp_path / 'easy.parquet')
    refs = ['君不见，黄河之水天上来！', '高堂明镜悲白发。']
    make_parquet(pq, refs)
    model = FakeModel(['君不见黄河之水天上来', '高堂明镜白发发'])
    out = str(tmp_path / 'full_page_easy.json')
    avg = ev.test_full_page(pq, out, model, None, None, dict(max_new_tokens=8), '读出图中所有文字。', True, False, False, 1.0, False,
                            batch_pages=batch_pages)
    rep = json.load(open(out, encoding='utf-8'))
    assert [d['imgPath'] for d in rep['detailed']] == ['p0.jpg', 'p1.jpg']
    d0, d1 = rep['detailed']
    assert d0['gt'] == '君不见黄河之水天上来' and d0['precision'] == d0['recall'] == d0['f1'] == 1.0 and d0['ned'] == 0.0
    assert d1['gt'] == '高堂明镜悲白发' and d1['output'] == '高堂明镜白发发'
    p, r, f = ev.calculate_metrics(list(d1['output']), list(d1['gt']))
    assert (d1['precision'], d1['recall'], d1['f1']) == (p, r, f) and abs(d1['ned'] - 2 / 7) < 1e-12
    assert set(rep['average']) == {'ave_precison', 'avg_recall', 'avg_f1', 'avg_ned'}       # upstream's key names
    assert abs(rep['average']['avg_ned'] - (0 + 2 / 7) / 2) < 1e-12 and avg == rep['average']
    assert len(model.calls) == (2 if batch_pages == 1 else 1)
    assert model.calls[0][1] == '读出图中所有文字。'


def test_full_page_report_sharded_over_ranks(tmp_path):
    """One process per GPU (BASELINE config 5): every rank scores its contiguous share of the pages, the rows are gathered in rank
    order and rank 0 writes the same report a single process writes."""
    pq = str(tmp_path / 'easy.parquet')
    refs = ['君不见，黄河之水天上来！', '高堂明镜悲白发。', '朝如青丝暮成雪', '人生得意须尽欢']
    answers = ['君不见黄河之水天上来', '高堂明镜白发发', '朝如青丝暮成雪', '人生得意']
    make_parquet(pq, refs)
    args = (None, None, dict(max_new_tokens=8), '读出图中所有文字。', True, False, False, 1.0, False)
    single = str(tmp_path / 'single.json')
    ev.test_full_page(pq, single, FakeModel(answers), *args, batch_pages=2)
    # world 3 (shares of 1, 1, 2 pages): ranks 2 and 1 first, their rows handed to rank 0's gather
    rows = {}
    for rank in (2, 1):
        lo, hi = 4 * rank // 3, 4 * (rank + 1) // 3
        out = str(tmp_path / f'rank{rank}.json')
        ev.test_full_page(pq, out, FakeModel(answers[lo:hi]), *args, batch_pages=2, shard=(rank, 3),
                          gather=lambda r, rank=rank: [[], rows.setdefault(rank, r) if rank == 1 else [], rows.setdefault(rank, r) if rank == 2 else []])
        assert not (tmp_path / f'rank{rank}.json').exists()             # only rank 0 writes
    out0 = str(tmp_path / 'rank0.json')
    avg = ev.test_full_page(pq, out0, FakeModel(answers[:1]), *args, batch_pages=2, shard=(0, 3), gather=lambda r: [r, rows[1], rows[2]])
    a, b = json.load(open(single, encoding='utf-8')), json.load(open(out0, encoding='utf-8'))
    assert a['detailed'] == b['detailed'] and [d['imgPath'] for d in b['detailed']] == ['p0.jpg', 'p1.jpg', 'p2.jpg', 'p3.jpg']
    for k in a['average']:
        assert abs(a['average'][k] - b['average'][k]) < 1e-12
    assert avg == b['average']


def test_compare_fp8_report(tmp_path):
    """--compare_fp8: the same pages scored twice (bf16, fp8 switches) -> per-page agreement and the F1 / NED deltas of the gate."""
    pq = str(tmp_path / 'easy.parquet')
    refs = ['君不见，黄河之水天上来！', '高堂明镜悲白发。', '朝如青丝暮成雪']
    make_parquet(pq, refs)
    args = (None, None, dict(max_new_tokens=8), '读出图中所有文字。', True, False, False, 1.0, False)
    a, b = str(tmp_path / 'bf16.json'), str(tmp_path / 'fp8.json')
    ev.test_full_page(pq, a, FakeModel(['君不见黄河之水天上来', '高堂明镜悲白发', '朝如青丝暮成雪']), *args, batch_pages=1)
    ev.test_full_page(pq, b, FakeModel(['君不见黄河之水天上来', '高堂明镜白发', '朝如青丝暮成霜']), *args, batch_pages=1)
    cmp = ev.compare_reports(json.load(open(a, encoding='utf-8'))['detailed'], json.load(open(b, encoding='utf-8'))['detailed'])
    assert cmp['pages'] == 3 and cmp['identical_outputs'] == 1
    assert abs(cmp['mean_ned_between_outputs'] - (0 + 1 / 7 + 1 / 7) / 3) < 1e-12
    assert cmp['avg_f1_bf16'] == 1.0 and cmp['avg_ned_bf16'] == 0.0
    assert cmp['delta_f1'] < 0 and abs(cmp['delta_ned'] - (1 / 7 + 1 / 7) / 3) < 1e-12


def test_main_refuses_to_score_without_t2s(monkeypatch):
    if ev.cc._cc is not None:
        pytest.skip('opencc is installed here')
    with pytest.raises(SystemExit, match='opencc'):
        ev.main(['--data', '/nonexistent'])


def test_choice_scoring_equals_the_reference():
    for c in GOLD['choice']:
        assert ev.evaluate_accuracy([c['response']], [tuple(c['answer'])]) == c['accuracy'], c
    rs, ans = [c['response'] for c in GOLD['choice']], [tuple(c['answer']) for c in GOLD['choice']]
    assert abs(ev.evaluate_accuracy(rs, ans) - sum(c['accuracy'] for c in GOLD['choice']) / len(rs)) < 1e-9


class TwoTurnModel:
    """chat_ocr stand-in that answers the transcription turn and then the task turn (history must be the first turn's)."""
    def __init__(self, answers):
        self.answers, self.calls = list(answers), []

    def chat_ocr(self, tokenizer, detect_model, image, prompt, generation_config, history=None, return_history=True, **kw):
        self.calls.append((image.size, prompt, history))
        if history is None:
            return '春眠不觉晓', [(prompt, '春眠不觉晓')]
        assert history == [(ev.FIRST_TURN, '春眠不觉晓')]
        return self.answers.pop(0), history + [(prompt, 'x')]


def make_task_parquet(path, rows):
    pd = pytest.importorskip('pandas')
    out = []
    for i, annot in enumerate(rows):
        buf = io.BytesIO()
        Image.new('RGB', (50 + i, 40), (255, 255, 255)).save(buf, format='PNG')
        out.append({'annotation': json.dumps(annot, ensure_ascii=False), 'image': {'bytes': buf.getvalue()}})
    pd.DataFrame(out).to_parquet(path)


def test_choice_and_reasoning_runs(tmp_path):
    q = '<image>\n这幅作品的作者是谁？\nA: 颜真卿\nB: 王羲之\nC: 柳公权'
    rows = [{'image': f'c{i}.jpg', 'conversations': [{'value': q}, {'value': 'B'}], 'content': '春眠不觉晓'} for i in range(4)]
    pq = str(tmp_path / 'author.parquet')
    make_task_parquet(pq, rows)
    assert ev.parse_choice(q, 'B') == ('B', '王羲之', '颜真卿', '柳公权')
    m = TwoTurnModel(['B', '颜真卿', 'A 王羲之', 'B'])
    acc, rep = ev.test_choice(pq, str(tmp_path / 'author.json'), m, None, None, {})          # the reference scores the first three samples
    assert rep['summary']['total_samples'] == 3 and abs(acc - 200 / 3) < 1e-9
    assert m.calls[1][1] == q.replace('<image>\n', '') + '\n只需要输出问题的答案，禁止输出其他内容！答案：'
    m = TwoTurnModel(['B', 'B', 'B', 'C'])
    acc, rep = ev.test_choice(pq, str(tmp_path / 'author_all.json'), m, None, None, {}, limit=0)
    assert rep['summary']['total_samples'] == 4 and acc == 75.0
    m = TwoTurnModel(['春眠 INTENT: 咏春'] * 4)
    out = ev.test_intent(pq, str(tmp_path / 'intent.json'), m, None, None, {}, True, False, False, 1.0, False)
    assert out[0] == {'imgPath': 'c0.jpg', 'chinese': '春眠 ', 'answer': ' 咏春', 'gt': 'B', 'calligraphy_content': '春眠不觉晓'}
    assert m.calls[1][1] == '<image>\n这幅作品的作者是谁？'                                   # the prompt up to its first option line
    assert 'calligraphy_content' not in ev.test_bilingual(pq, str(tmp_path / 'bi.json'), TwoTurnModel(['x'] * 4), None, None, {}, True, False, False, 1.0, False)[0]


def test_region_wise_run(tmp_path):
    rows = [{'imagePath': 'r0.jpg', 'region': [[5, 4], [30, 20]], 'answer': '高堂，明镜'}, {'imagePath': 'r1.jpg', 'region': [[0, 0], [10, 10]], 'answer': '白发'}]
    pq = str(tmp_path / 'region.parquet')
    make_task_parquet(pq, rows)
    m = FakeModel(['高堂明镜', '白头'])
    avg = ev.test_region_wise(pq, str(tmp_path / 'region.json'), m, None, None, {}, '读出图中区域所有文字。', True, False, False, 1.0, False)
    rep = json.load(open(str(tmp_path / 'region.json'), encoding='utf-8'))
    assert m.calls[0][0] == (25, 16) and m.calls[1][0] == (10, 10)                           # the crops, not the pages
    assert rep['detailed'][0]['f1'] == 1.0 and rep['detailed'][1]['gt'] == '白发' and abs(avg['avg_f1'] - rep['average']['avg_f1']) < 1e-12
