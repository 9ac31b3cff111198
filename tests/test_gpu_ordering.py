"""Ordering front end, GPU side: OrderFormer on the HIP path against the oracle model, and the whole
detections -> reading order pipeline against the reference's results (tests/golden/ordering_vectors.json)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from callireader_amd import ordering, synthetic            # noqa: E402
from oracle import orderformer as oracle_of                # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'ordering_vectors.json')))
PAGES = [k for k in GOLD if not k.startswith('_')]


@pytest.fixture(scope='module')
def model():
    from callireader_amd.config import ModelDims
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    m = InternVLChatModel.from_synthetic(ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1), seed=0, device=0)
    m.load_orderformer(synthetic.make_orderformer_state_dict(seed=GOLD['_model']['state_dict_seed']))
    return m


@pytest.fixture(scope='module')
def cpu_scorer():
    return oracle_of.CpuScorer(synthetic.make_orderformer_state_dict(seed=GOLD['_model']['state_dict_seed']))


def test_scores_match_oracle(model, cpu_scorer):
    """bf16 through 4 post-norm layers: the HIP path keeps the eager module's rounding points but sums in a different
    order, so scores agree to a few bf16 steps of their magnitude; padded pages and a batch of pages are covered."""
    g = torch.Generator().manual_seed(21)
    x = torch.zeros(5, 50, 4)
    for b, n in enumerate((50, 33, 7, 1, 50)):
        x[b, :n] = torch.rand(n, 4, generator=g)
    x = x.to(torch.bfloat16)
    got = model.engine.orderformer(x).cpu()
    want = cpu_scorer.orderformer(x)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (5, 50)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 0.04 * scale, (float((got - want).abs().max()), scale)
    # a page's scores do not depend on what it is batched with
    alone = model.engine.orderformer(x[1:2]).cpu()
    assert torch.equal(alone[0], got[1])


@pytest.mark.parametrize('name', PAGES)
def test_reading_order_matches_reference(name, model, cpu_scorer):
    """Whole front end on the GPU scorer.  The order is a sort of scores, so it is compared wherever the oracle's own
    neighbouring scores are further apart than the two implementations can differ; closer calls are only required
    to stay a permutation of the same boxes."""
    g = GOLD[name]
    got = ordering.sort_boxes(g['raw_boxes'], g['width'], g['height'], model.sorter)
    assert sorted(got) == sorted(g['final'])
    page = ordering.chars_to_columns([[list(b[0]), list(b[1])] for b in ordering.clean_detections(g['raw_boxes'], g['width'], g['height'])],
                                     g['width'], g['height'])
    n = len(page['shapes'])
    probe = ordering.OrderFormer(_Recorder(cpu_scorer), max_nums=50)
    probe.predict(json.loads(json.dumps(page)))
    s = probe.engine.last[0, :n].sort().values
    gap = float((s[1:] - s[:-1]).min()) if n > 1 else 1.0
    if gap > 0.08 * float(probe.engine.last.abs().max()):
        assert got == g['final']


class _Recorder:
    def __init__(self, inner):
        self.inner, self.last = inner, None

    def orderformer(self, x):
        self.last = self.inner.orderformer(x)
        return self.last


def test_calli_align_runs_the_front_end_on_raw_detections(model):
    """calli_align with a sorter loaded: the detector hands over unordered boxes; tiles come out in reading order."""
    from PIL import Image
    import numpy as np
    g = GOLD['three_columns']
    img = Image.fromarray(np.full((g['height'], g['width'], 3), 255, dtype=np.uint8))
    seen = {}

    def detector(arr):
        seen['shape'] = arr.shape
        return g['raw_boxes']
    emb, idx = model.calli_align(img, detector)
    assert seen['shape'] == (g['height'], g['width'], 3)
    n = len(ordering.clean_detections(g['raw_boxes'], g['width'], g['height']))
    assert emb.shape == (3 * n, model.dims.llm_hidden) and idx.shape[0] == n
