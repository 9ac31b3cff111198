#!/usr/bin/env python3
"""Golden vectors for the ordering front end (everything after the detector), produced by the reference's
own code in the build container:
  * utils/utils.py: calculate_iou, merge_boxes (imported);
  * the closures inside InternVLChatModel.calli_align -- iterative_only_boxes (box clean-up after the detector),
    char2col_with_kmeans, sort_boxes -- are not importable, so their definitions are compiled out of the reference's
    source AT GENERATION TIME (nothing of it is stored) and driven with a fake detector that returns the given boxes;
  * models/model.py: OrderFormer (predict / _decode / postprocess) with seeded bf16 weights on the CPU.
Output: tests/golden/ordering_vectors.json (inputs and expected outputs only)."""
import ast
import inspect
import json
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs  # noqa: E402

install_stubs()
tvt = sys.modules['torchvision.transforms']
tvt.Compose = lambda x: x
tvt.Lambda = tvt.Resize = tvt.ToTensor = tvt.Normalize = lambda *a, **k: None
sys.modules['torchvision'].transforms = tvt
for name in ['tqdm']:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)

from sklearn.cluster import KMeans  # noqa: E402
from utils.utils import calculate_iou, merge_boxes  # noqa: E402
import models.model as ref_model  # noqa: E402
import InternVL.modeling_internvl_chat as ref_chat  # noqa: E402
from callireader_amd import synthetic  # noqa: E402


def nested_functions(outer, names):
    """Compile the nested defs `names` of `outer` into a namespace that provides what they close over."""
    src = textwrap.dedent(inspect.getsource(outer))
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            found[node.name] = node
    ns = {'np': np, 'KMeans': KMeans, 'merge_boxes': merge_boxes, 'calculate_iou': calculate_iou,
          'coord_transform': ref_chat.coord_transform}
    for name in names:
        mod = ast.Module(body=[found[name]], type_ignores=[])
        exec(compile(mod, f'<reference:{name}>', 'exec'), ns)
    return ns


class FakeImage:
    def __init__(self, w, h):
        self.shape = (h, w, 3)


class FakeBox:
    def __init__(self, b):
        self.xyxy = torch.tensor([b], dtype=torch.float32)


class FakeResult:
    def __init__(self, boxes):
        self.boxes = [FakeBox(b) for b in boxes]


def run_reference(raw_boxes, w, h, sorter):
    ns = nested_functions(ref_chat.InternVLChatModel.calli_align, ['iterative_only_boxes', 'char2col_with_kmeans', 'sort_boxes'])
    img = FakeImage(w, h)
    ns['dynamic_read'] = lambda path, mode='c': img
    ns['most_frequent_rgb_fast'] = lambda a: (255, 255, 255)
    ns['mask_area'] = lambda a, b, c: a
    np_array = np.array
    # np.array(image) inside iterative_only_boxes: hand the fake image through untouched
    class NP:
        def __getattr__(self, k):
            return getattr(np, k)
        def array(self, x, *a, **k):
            return x if isinstance(x, FakeImage) else np_array(x, *a, **k)
    ns['np'] = NP()
    detector = lambda image, verbose=False: [FakeResult(raw_boxes)]
    cleaned = ns['iterative_only_boxes'](detector, 'unused.jpg')
    cols = ns['char2col_with_kmeans']('unused.jpg', [[list(b[0]), list(b[1])] for b in cleaned])
    ordered_cols = sorter.predict(json.loads(json.dumps(cols)), 'unused.jpg')
    final = ns['sort_boxes']('unused.jpg', detector, sorter)
    return cleaned, cols, ordered_cols, final


def synth_page(rng, w, h, n_cols, per_col, seal=False, dup=0):
    """vertical text, columns right to left; optional inscription column of small characters; `dup` near-duplicates"""
    boxes = []
    cw = w / (n_cols + 1.5)
    ch = (h * 0.9) / per_col
    for c in range(n_cols):
        x0 = w - (c + 1) * cw - cw * 0.2
        for r in range(per_col - (c == n_cols - 1) * 2):
            jx, jy = rng.uniform(-0.05, 0.05, 2)
            bw, bh = cw * rng.uniform(0.7, 0.85), ch * rng.uniform(0.75, 0.9)
            x1, y1 = x0 + jx * cw, h * 0.05 + r * ch + jy * ch
            boxes.append([x1, y1, x1 + bw, y1 + bh])
    if seal:                     # an inscription: two columns of small characters at the left edge
        sch = ch * 0.22
        for c in range(2):
            sx = w * 0.03 + c * cw * 0.3
            for r in range(12):
                boxes.append([sx + rng.uniform(-1, 1), h * 0.3 + r * sch, sx + cw * 0.17, h * 0.3 + r * sch + sch * 0.85])
    for i in range(dup):
        b = boxes[int(rng.integers(len(boxes)))]
        boxes.append([b[0] + 1, b[1] + 1, b[2] + 1.5, b[3] + 1])
    boxes = [[float(int(v)) for v in b] for b in boxes]
    order = rng.permutation(len(boxes))
    return [boxes[i] for i in order]


def main():
    torch.manual_seed(0)
    torch.set_default_dtype(torch.bfloat16)          # the reference builds the sorter inside a bf16 from_pretrained
    sorter = ref_model.OrderFormer(max_nums=50, input_dim=4, model_dim=256, num_heads=8, num_layers=4, output_dim=1,
                                   device=torch.device('cpu'), label_name='turn', norm=False)
    torch.set_default_dtype(torch.float32)
    sd = synthetic.make_orderformer_state_dict(seed=11)
    sorter.model.load_state_dict({k: v for k, v in sd.items()})
    sorter.model.eval()

    cases = {}
    ex = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'example0_boxes.json')))
    rng = np.random.default_rng(3)
    W0, H0 = ex['imageWidth'], ex['imageHeight']          # labelme-style fixture, points normalised to [0, 1]
    raw = [[float(int(sh['points'][0][0] * W0)), float(int(sh['points'][0][1] * H0)), float(int(sh['points'][1][0] * W0)),
            float(int(sh['points'][1][1] * H0))] for sh in ex['shapes']]
    raw = [raw[i] for i in rng.permutation(len(raw))]
    inputs = {'example0': (raw, W0, H0)}
    inputs['three_columns'] = (synth_page(rng, 900, 1400, 3, 9), 900, 1400)
    inputs['with_inscription'] = (synth_page(rng, 1200, 1600, 4, 8, seal=True), 1200, 1600)
    inputs['duplicates'] = (synth_page(rng, 1000, 1000, 2, 6, dup=5), 1000, 1000)
    inputs['single_column'] = (synth_page(rng, 400, 1500, 1, 10), 400, 1500)
    with torch.no_grad():
        for name, (boxes, w, h) in inputs.items():
            cleaned, cols, ordered_cols, final = run_reference(boxes, w, h, sorter)
            # raw model scores for the OrderFormer parity test
            cases[name] = {'width': w, 'height': h, 'raw_boxes': boxes, 'cleaned': cleaned, 'columns': cols,
                           'ordered_columns': {str(k): v for k, v in ordered_cols.items()}, 'final': final}
        # model-level vectors: padded inputs -> scores
        g = torch.Generator().manual_seed(5)
        x = torch.zeros(3, 50, 4)
        for b, n in enumerate((50, 17, 1)):
            x[b, :n] = torch.rand(n, 4, generator=g)
        xb = x.to(torch.bfloat16)
        y = sorter.model(xb)
    cases['_model'] = {'x': xb.float().tolist(), 'y': y.float().reshape(3, 50).tolist(), 'state_dict_seed': 11}
    # unit vectors for the two helpers
    rngu = np.random.default_rng(8)
    ious = []
    for _ in range(12):
        a = sorted(rngu.uniform(0, 100, 2).tolist()) + sorted(rngu.uniform(0, 100, 2).tolist())
        b = sorted(rngu.uniform(0, 100, 2).tolist()) + sorted(rngu.uniform(0, 100, 2).tolist())
        A, Bx = [a[0], a[2], a[1], a[3]], [b[0], b[2], b[1], b[3]]
        ious.append({'a': A, 'b': Bx, 'iou': calculate_iou(A, Bx), 'iou_min': calculate_iou(A, Bx, mini=True)})
    cases['_iou'] = ious
    json.dump(cases, open(os.path.join(ROOT, 'tests', 'golden', 'ordering_vectors.json'), 'w'), indent=0)
    print('ok', {k: (len(v['raw_boxes']), len(v['cleaned']), len(v['columns']['shapes']), len(v['final']),
                     v['columns']['boxes2class'] is not None) for k, v in cases.items() if not k.startswith('_')})


if __name__ == '__main__':
    main()
