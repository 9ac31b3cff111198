"""Ordering front end, CPU side: the host logic of callireader_amd/ordering.py against vectors produced
by the reference's own functions (scripts/make_golden_ordering.py), with the oracle model standing in for the GPU
scorer; and the oracle model against the reference's scores."""
import json
import os

import numpy as np
import pytest
import torch

from callireader_amd import ordering, synthetic
from oracle import orderformer as oracle_of

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'ordering_vectors.json')))
PAGES = [k for k in GOLD if not k.startswith('_')]


@pytest.fixture(scope='module')
def scorer():
    return oracle_of.CpuScorer(synthetic.make_orderformer_state_dict(seed=GOLD['_model']['state_dict_seed']))


def test_iou_vectors():
    for v in GOLD['_iou']:
        assert ordering.box_iou(v['a'], v['b']) == v['iou']
        assert ordering.box_iou(v['a'], v['b'], over_min=True) == v['iou_min']


def test_oracle_model_is_the_reference_model(scorer):
    x = torch.tensor(GOLD['_model']['x'])
    y = scorer.orderformer(x)
    assert torch.equal(y, torch.tensor(GOLD['_model']['y']))              # same torch modules: zero difference


@pytest.mark.parametrize('name', PAGES)
def test_cleaning_and_columns_match_reference(name):
    g = GOLD[name]
    cleaned = ordering.clean_detections(g['raw_boxes'], g['width'], g['height'])
    assert cleaned == g['cleaned']
    page = ordering.chars_to_columns([[list(b[0]), list(b[1])] for b in cleaned], g['width'], g['height'])
    want = g['columns']
    assert page['shapes'] == want['shapes']
    for key in ('boxes2class', 'col2class'):
        got = page[key]
        assert (got is None) == (want[key] is None)
        if got is not None:
            assert {str(k): v for k, v in got.items()} == want[key]


def test_the_inscription_case_takes_the_kmeans_branch():
    assert GOLD['with_inscription']['columns']['boxes2class'] is not None
    assert GOLD['duplicates']['cleaned'] != [[[b[0], b[1]], [b[2], b[3]]] for b in GOLD['duplicates']['raw_boxes']]


@pytest.mark.parametrize('name', PAGES)
def test_order_matches_reference(name, scorer):
    g = GOLD[name]
    sorter = ordering.OrderFormer(scorer, max_nums=50)
    cols = sorter.predict(json.loads(json.dumps(g['columns'])))
    assert {str(k): v for k, v in cols.items()} == g['ordered_columns']
    assert ordering.sort_boxes(g['raw_boxes'], g['width'], g['height'], sorter) == g['final']


def test_detect_all_repaints_and_repeats():
    import numpy as np
    img = np.full((40, 60, 3), 200, dtype=np.uint8)
    img[5:10, 5:10] = 0
    calls = []

    def detector(a):
        calls.append(a.copy())
        return [[1, 1, 4, 4], [5, 5, 10, 10]] if len(calls) == 1 else [[20, 20, 30, 30]]
    out = ordering.detect_all(detector, img, max_per_pass=1)
    assert out == [[1, 1, 4, 4], [5, 5, 10, 10], [20, 20, 30, 30]]
    assert (calls[1][5:10, 5:10] == 200).all() and img[5, 5, 0] == 0          # painted over with the dominant colour, on a copy


def test_too_many_columns_is_an_error(scorer):
    page = {'imageWidth': 100, 'imageHeight': 100, 'shapes': [{'points': [[i, 0], [i + 1, 5]]} for i in range(51)]}
    with pytest.raises(ValueError):
        ordering.OrderFormer(scorer, max_nums=50).predict(page)


def test_clean_detections_fast_path_is_the_sweep_and_the_dominant_colour_keeps_its_tie_rule():
    """Round 6 (host time per page on the thread that feeds the GPU): clean_detections answers from one vectorised IoU matrix when no pair exceeds the threshold and runs the
    reference's sweep (:369-392, removal by value) otherwise -- against the sweep alone on 400 random box sets, a third of them with near-duplicates; most_frequent_rgb by
    np.unique keeps bincount + argmax's tie rule (the smallest packed colour) and detect_all only computes it when a pass returns more than 250 boxes."""
    import random
    from callireader_amd import ordering

    def sweep(boxes, width, height, thr=0.8):
        out = [[[max(b[0], 0), max(b[1], 0)], [min(b[2], width), min(b[3], height)]] for b in boxes]
        i, n = 0, len(out)
        while i < n:
            keep, j = out[i], 0
            while j < n:
                if j != i and ordering.box_iou(ordering._flat(keep), ordering._flat(out[j])) > thr:
                    out.remove(out[j])
                    if j < i:
                        i -= 1
                    n -= 1
                    j -= 1
                j += 1
            i += 1
        return out
    rng = random.Random(3)
    for t in range(400):
        bs = []
        for _ in range(rng.randint(1, 60)):
            x1, y1 = rng.randint(-5, 700), rng.randint(-5, 1900)
            bs.append([x1, y1, x1 + rng.randint(5, 120), y1 + rng.randint(5, 120)])
        if t % 3 == 0:
            for _ in range(rng.randint(1, 5)):
                b = rng.choice(bs)
                bs.append([b[0] + rng.randint(0, 3), b[1] + rng.randint(0, 3), b[2] + rng.randint(0, 3), b[3]])
        assert ordering.clean_detections(bs, 788, 2000) == sweep(bs, 788, 2000), t
    g = np.random.default_rng(0)
    for _ in range(8):
        a = (g.integers(0, 4, (40, 50, 3)) * 60).astype(np.uint8)
        flat = a.reshape(-1, 3).astype(np.int64)
        top = int(np.argmax(np.bincount((flat[:, 0] << 16) | (flat[:, 1] << 8) | flat[:, 2])))
        assert ordering.most_frequent_rgb(a) == ((top >> 16) & 255, (top >> 8) & 255, top & 255)
    calls = []
    orig = ordering.most_frequent_rgb
    ordering.most_frequent_rgb = lambda im: calls.append(1) or orig(im)
    try:
        page = np.full((300, 300, 3), 200, dtype=np.uint8)
        few = [[10 * k, 10, 10 * k + 8, 40] for k in range(20)]
        assert ordering.detect_all(lambda arr, verbose=False: few, page) == few and not calls
        many = [[(k % 28) * 10, (k // 28) * 10, (k % 28) * 10 + 8, (k // 28) * 10 + 8] for k in range(260)]
        state = {'n': 0}

        def det(arr, verbose=False):
            state['n'] += 1
            return many if state['n'] == 1 else few
        assert ordering.detect_all(det, page) == many + few and len(calls) == 1
    finally:
        ordering.most_frequent_rgb = orig
