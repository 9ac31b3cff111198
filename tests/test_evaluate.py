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
