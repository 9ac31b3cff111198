"""Ordering front end, CPU side: the host logic of callireader_amd/ordering.py against vectors produced
by the reference's own functions (scripts/make_golden_ordering.py), with the oracle model standing in for the GPU
scorer; and the oracle model against the reference's scores."""
import json
import os

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
