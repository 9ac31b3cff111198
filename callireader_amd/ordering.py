"""Ordering front end of `chat_ocr`: everything between the character detector and the tile path.

Reference behaviour (InternVL/modeling_internvl_chat.py:346-556, utils/utils.py:20-41,230-331, models/model.py:206-526):
  detector boxes -> clip + drop near-duplicates (IoU > 0.8)                      `clean_detections`
                 -> columns: greedy vertical merge, optional 2-means split by
                    area into body text / inscription                            `merge_columns`, `split_by_area`, `chars_to_columns`
                 -> OrderFormer scores the columns, argsort = reading order,
                    then a local three-box fix-up                                `OrderFormer.predict`, `.postprocess`
                 -> characters of each column (IoU-over-min >= 0.8), top to
                    bottom                                                       `sort_boxes`
The detector itself (ultralytics YOLOv10 weights) is third-party and not part of this package: pass any callable
`image (H, W, 3 uint8) -> [[x1, y1, x2, y2], ...]`.

Every function here is pinned to the reference's own code on five pages (tests/golden/ordering_vectors.json, written
by scripts/make_golden_ordering.py).  The quirks of the reference are kept on purpose where they decide the result
(list removal by value, the index bookkeeping of its in-place merge loops, an operator-precedence slip in the
inscription test): a drop-in has to order real pages the way the reference does.

OrderFormer runs on the GPU through the C ABI (`cr_orderformer`, csrc/orderformer.hip); the clustering uses scikit-learn's
KMeans(n_clusters=2, random_state=0) exactly like the reference.
"""
import numpy as np
import torch


# ---- geometry ------------------------------------------------------------------------------------------------------
def box_iou(a, b, over_min=False):
    """utils/utils.py:20-41.  Boxes as [x1, y1, x2, y2]; `over_min` divides by the smaller area instead of the union."""
    iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (b[2] - b[0]) * (b[3] - b[1])
    return inter / min(area_a, area_b) if over_min else inter / (area_a + area_b - inter)


def _flat(pair):
    return [pair[0][0], pair[0][1], pair[1][0], pair[1][1]]


def _overlap_1d(a0, a1, b0, b1):
    lo, hi = max(a0, b0), min(a1, b1)
    return hi - lo if lo < hi else 0


def _gap_1d(a0, a1, b0, b1):
    return 0 if _overlap_1d(a0, a1, b0, b1) > 0 else min(abs(a0 - b1), abs(b0 - a1))


def most_frequent_rgb(image):
    """modeling_internvl_chat.py:98-115 (the colour detected boxes are painted over with between detector passes)."""
    flat = np.asarray(image).reshape(-1, 3).astype(np.int32)
    packed = (flat[:, 0] << 16) | (flat[:, 1] << 8) | flat[:, 2]
    values, counts = np.unique(packed, return_counts=True)       # ascending values: arg-max of the counts = the SMALLEST packed colour among ties, as bincount + argmax gives
    top = int(values[int(np.argmax(counts))])                    # (a 2^24-bin bincount of a 1.6-megapixel page cost 430 ms)
    return ((top >> 16) & 255, (top >> 8) & 255, top & 255)


def _takes_verbose(detector):
    if hasattr(detector, 'predict'):
        return True
    import inspect
    try:
        params = inspect.signature(detector).parameters
    except (TypeError, ValueError):
        return False
    return 'verbose' in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())


def acquire_boxes(detector, image, sorter):
    """The box acquisition of calli_align (modeling_internvl_chat.py:346-394,558), shared by chat_ocr and the batched page
    path: with a sorter, repeated detection passes + the reference's ordering front end; without one, a single detector pass
    whose order is taken as the reading order."""
    if not callable(detector):
        raise NotImplementedError('calli_align needs a detector: pass the ultralytics YOLO object the reference '
                                  'uses (inference.py:98), any callable image -> boxes, or boxes=[(x1,y1,x2,y2),...]')
    arr = np.array(image)
    if sorter is not None:
        return sort_boxes(detect_all(detector, arr), image.width, image.height, sorter)
    return run_detector(detector, arr)


def run_detector(detector, image):
    """One detection pass -> [[x1, y1, x2, y2], ...] (ints, truncated as the reference does).

    `detector` is what the reference hands to chat_ocr (inference.py:37-42,98): an ultralytics `YOLO` object, used as
    `detector(image_array, verbose=False)[0].boxes[i].xyxy` (modeling_internvl_chat.py:356-362) -- or any callable
    `image_array -> iterable of (x1, y1, x2, y2, ...)` (tests, other detectors)."""
    # the ultralytics call shape is chosen by looking at the object (a YOLO model has .predict), not by catching TypeError:
    # a TypeError raised INSIDE a real detector must surface, not trigger a silent second run
    res = detector(image, verbose=False) if _takes_verbose(detector) else detector(image)
    first = res[0] if isinstance(res, (list, tuple)) and len(res) and hasattr(res[0], 'boxes') else None
    if first is None and hasattr(res, 'boxes'):
        first = res
    if first is not None:
        out = []
        for box in first.boxes:
            xyxy = box.xyxy
            xyxy = xyxy.squeeze().tolist() if hasattr(xyxy, 'squeeze') else list(xyxy)
            out.append([int(xyxy[0]), int(xyxy[1]), int(xyxy[2]), int(xyxy[3])])
        return out
    return [[int(v) for v in b[:4]] for b in res]


def detect_all(detector, image, max_per_pass=250):
    """:346-368.  Detectors cap their output, so while a pass returns more than `max_per_pass` boxes the found ones are
    painted over with the page's dominant colour and the detector runs again.  Coordinates are truncated to int."""
    image = np.array(image)
    colour = None                                  # the reference computes it before the first pass (:355); it is only ever used after a pass of > max_per_pass boxes,
    found = []                                     # and the page is still unpainted then: the same colour, without 100+ ms of host work on every ordinary page
    while True:
        batch = run_detector(detector, image)
        found.extend(batch)
        if len(batch) <= max_per_pass:
            return found
        if colour is None:
            colour = most_frequent_rgb(image)
        for x1, y1, x2, y2 in batch:
            image[y1:y2, x1:x2] = colour


def clean_detections(boxes, width, height, iou_thr=0.8):
    """:369-392.  Clip to the page, then drop every box that overlaps an earlier-kept one with IoU > `iou_thr`.
    Returns [[x1, y1], [x2, y2]] pairs in the reference's order."""
    out = [[[max(b[0], 0), max(b[1], 0)], [min(b[2], width), min(b[3], height)]] for b in boxes]
    if len(out) > 1:
        # The usual page has no such pair at all, and the sweep below is n^2 Python-level IoUs (18 ms for the example page's 96 boxes, on the thread that feeds the GPU):
        # all IoUs at once first -- integer coordinates, so the float64 quotients are the sweep's own -- and the sweep only when some pair exceeds the threshold
        # (or a box is empty: the sweep's division then fails as it always did).
        a = np.asarray([_flat(b) for b in out], dtype=np.float64)
        iw = np.clip(np.minimum(a[:, None, 2], a[None, :, 2]) - np.maximum(a[:, None, 0], a[None, :, 0]), 0, None)
        ih = np.clip(np.minimum(a[:, None, 3], a[None, :, 3]) - np.maximum(a[:, None, 1], a[None, :, 1]), 0, None)
        inter = iw * ih
        area = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        union = area[:, None] + area[None, :] - inter
        np.fill_diagonal(union, 1.0)
        np.fill_diagonal(inter, 0.0)
        if (union > 0).all() and not (inter / union > iou_thr).any():
            return out
    i, n = 0, len(out)
    while i < n:
        keep = out[i]                                    # fixed for the whole sweep, as in the reference
        j = 0
        while j < n:
            if j != i and box_iou(_flat(keep), _flat(out[j])) > iou_thr:
                out.remove(out[j])                       # by value: the first equal box goes
                if j < i:
                    i -= 1
                n -= 1
                j -= 1
            j += 1
        i += 1
    return out


def merge_columns(boxes, thresx=0.7, thresy=2):
    """utils/utils.py:273-331.  Boxes whose x-ranges overlap by more than `thresx` of the narrower one and whose
    vertical gap is under `thresy` mean heights are united, repeatedly (at most 10 sweeps), starting from the boxes
    sorted by their y-centre.  Input and output are [[x1, y1], [x2, y2]] pairs."""
    cols = sorted(boxes, key=lambda b: (b[0][1] + b[1][1]) / 2)
    before = len(cols)
    for _ in range(10):
        n = len(cols)
        if n == 0:
            break
        i = 0
        while i < n:
            j = 0
            while j < n:
                if j == i:
                    j += 1
                    continue
                a, b = cols[i], cols[j]
                n = len(cols)
                wa, wb = abs(a[0][0] - a[1][0]), abs(b[0][0] - b[1][0])
                ha, hb = abs(a[0][1] - a[1][1]), abs(b[0][1] - b[1][1])
                x_rate = _overlap_1d(a[0][0], a[1][0], b[0][0], b[1][0]) / min(wa, wb)
                y_rate = _gap_1d(a[0][1], a[1][1], b[0][1], b[1][1]) / ((ha + hb) / 2)
                if x_rate > thresx and y_rate < thresy:
                    cols[i] = [[min(a[0][0], b[0][0]), min(a[0][1], b[0][1])], [max(a[1][0], b[1][0]), max(a[1][1], b[1][1])]]
                    cols.remove(b)                       # by value
                    if j < i:
                        i -= 1
                    n -= 1
                    j -= 1
                j += 1
            i += 1
        if len(cols) == before:
            break
        before = len(cols)
    return cols


def split_by_area(norm_boxes):
    """:397-469 (`kmeans_boxes`).  2-means on the box areas separates body text from the inscription; boxes of the
    small-character group that are wide or tall enough, or that still merge into columns with their peers, are handed
    back to the body group.  Returns (group_0, group_1) in the reference's order."""
    from sklearn.cluster import KMeans
    w_of = lambda b: b[1][0] - b[0][0]
    h_of = lambda b: b[1][1] - b[0][1]
    areas = np.array([w_of(b) * h_of(b) for b in norm_boxes]).reshape(-1, 1)
    labels = KMeans(n_clusters=2, random_state=0).fit(areas).labels_
    groups = [[b for b, l in zip(norm_boxes, labels) if l == 0], [b for b, l in zip(norm_boxes, labels) if l != 0]]
    groups = [sorted(g, key=w_of, reverse=True) for g in groups]
    widest = [w_of(g[0]) for g in groups]
    if widest[0] == widest[1]:
        return groups[0], groups[1]
    body, small = (1, 0) if widest[1] > widest[0] else (0, 1)
    mean_h = np.array([h_of(b) for b in groups[body]]).mean()
    thr_w, thr_h = w_of(groups[body][-1]), 0.8 * mean_h
    rest = []
    for b in groups[small]:
        # (min area / width) * height, as the reference's expression parses
        inscription_like = areas.min() / w_of(b) * h_of(b) <= 1 / 5 and areas.mean() / (w_of(b) * h_of(b)) <= 1.3
        if w_of(b) >= thr_w or h_of(b) >= thr_h or inscription_like:
            groups[body].append(b)
        else:
            rest.append(b)
    merged = merge_columns(rest.copy())
    kept = []
    for b in rest:
        if b in merged:                                  # a box that merged with nothing is a column of its own
            groups[body].append(b)
        else:
            kept.append(b)
    groups[small] = kept
    return groups[0], groups[1]


def _ints(v):
    return [[int(v[0][0]), int(v[0][1])], [int(v[1][0]), int(v[1][1])]] if len(v) == 2 else [int(x) for x in v]


def chars_to_columns(boxes, width, height):
    """:471-514 (`char2col_with_kmeans`): labelme-style dict of column boxes, plus the two classes when the page's box
    areas are spread enough (coefficient of variation > 0.66 and smallest / mean <= 1/8) to call for the split."""
    norm = [[[b[0][0] / width, b[0][1] / height], [b[1][0] / width, b[1][1] / height]] for b in boxes]
    areas = np.array([(b[0][0] - b[1][0]) * (b[0][1] - b[1][1]) for b in norm])
    boxes2class = col2class = None
    if np.std(areas) / np.mean(areas) > 0.66 and areas.min() / areas.mean() <= 1 / 8:
        g0, g1 = split_by_area(norm)
        scale = lambda g: [[[b[0][0] * width, b[0][1] * height], [b[1][0] * width, b[1][1] * height]] for b in g]
        g0, g1 = scale(g0), scale(g1)
        c0, c1 = merge_columns(g0.copy()), merge_columns(g1.copy())
        columns = c0 + c1
        boxes2class = {1: [_ints(b) for b in g0], 2: [_ints(b) for b in g1]}
        col2class = {1: [_ints(b) for b in c0], 2: [_ints(b) for b in c1]}
    else:
        columns = merge_columns(boxes.copy())
    return {'imageHeight': height, 'imageWidth': width, 'shapes': [{'points': _ints(c)} for c in columns],
            'boxes2class': boxes2class, 'col2class': col2class}


# ---- OrderFormer ----------------------------------------------------------------------------------------------------
class OrderFormer:
    """models/model.py:235-526, inference side.  A 4-layer post-norm encoder (d = 256, 8 heads, ReLU FFN 2048) scores
    up to `max_nums` boxes; ascending score = reading order.  The model runs through `Engine.orderformer` (HIP)."""

    def __init__(self, engine, max_nums=50, input_dim=4):
        self.engine = engine
        self.max_nums = max_nums
        self.input_dim = input_dim

    @classmethod
    def from_state_dict(cls, engine, state_dict, max_nums=50):
        """`state_dict`: the reference's `Transformer` keys (params/orderformer.pth); the template `encoder_layer.*`
        entries are ignored, as the reference's forward never runs them."""
        for k, v in state_dict.items():
            if k.startswith('encoder_layer.'):
                continue
            engine.load_weight('orderformer.' + k, v.to(torch.bfloat16))      # the reference runs the sorter in bf16
        return cls(engine, max_nums=max_nums)

    @staticmethod
    def _sort_key(entry):
        c = entry[0]
        return ((c[0] + c[2]) / 2) ** 2 + ((c[1] + c[3]) / 2) ** 2

    def scores(self, batch):
        """batch: (B, max_nums, 4) float -> (B, max_nums) fp32 scores (inputs are rounded to bf16 like the reference's)."""
        x = torch.as_tensor(batch, dtype=torch.float32).to(torch.bfloat16)
        return self.engine.orderformer(x)

    @staticmethod
    def decode(scores, n):
        """:325-332: rank (1-based) of each of the first n scores."""
        order = torch.argsort(scores.reshape(1, -1)[:, :n], dim=1)
        return torch.argsort(order, dim=1) + 1

    def predict(self, page):
        """:419-484.  `page`: labelme-style dict (`chars_to_columns`' output) -> {rank: [x1, y1, x2, y2]} sorted by rank."""
        w, h = page['imageWidth'], page['imageHeight']
        pts = [s['points'] for s in page['shapes']]
        if len(pts) > self.max_nums:
            raise ValueError(f'{len(pts)} boxes, OrderFormer takes at most {self.max_nums}')
        xs = np.array([v for p in pts for v in (p[0][0] / w, p[1][0] / w)])
        ys = np.array([v for p in pts for v in (p[0][1] / h, p[1][1] / h)])
        xs, ys = xs - xs.min(), ys - ys.min()                       # translation invariance
        entries = [[[xs[2 * i], ys[2 * i], xs[2 * i + 1], ys[2 * i + 1]], _flat(p)] for i, p in enumerate(pts)]
        entries = sorted(entries, key=self._sort_key)               # canonical input order: distance of the centre from the origin
        flat = [v for e in entries for v in e[0]] + [0] * self.input_dim * (self.max_nums - len(entries))
        x = torch.tensor(flat, dtype=torch.bfloat16).reshape(1, self.max_nums, self.input_dim)
        ranks = self.decode(self.engine.orderformer(x).cpu(), len(entries)).squeeze().tolist()
        if isinstance(ranks, int):
            ranks = [ranks]
        ranked = dict(sorted({r: e[1] for r, e in zip(ranks, entries)}.items()))
        return dict(sorted(self.postprocess(ranked, w, h).items()))

    @staticmethod
    def postprocess(results, width, height):
        """:488-526.  Sliding window over the ranked boxes: three boxes of similar size on one text line are put right
        to left; otherwise the window is reversed -- exactly the reference's rule, side effects included."""
        def window_order(b1, b2, b3):
            ws = [b[2] - b[0] for b in (b1, b2, b3)]
            hs = [b[3] - b[1] for b in (b1, b2, b3)]
            cx = [(b[0] + b[2]) / 2 for b in (b1, b2, b3)]
            cy = [(b[1] + b[3]) / 2 for b in (b1, b2, b3)]
            area = [w_ * h_ for w_, h_ in zip(ws, hs)]
            same_line = max(abs(cy[0] - cy[1]), abs(cy[0] - cy[2]), abs(cy[1] - cy[2])) < min(hs) and min(area) / max(area) > 0.7
            key = cx if same_line else [3, 2, 1]
            return [i for i, _ in sorted(enumerate(key), key=lambda t: t[1], reverse=True)]
        boxes = [[b[0] / width, b[1] / height, b[2] / width, b[3] / height] for b in results.values()]
        for i in range(len(results) - 2):
            o = window_order(boxes[i], boxes[i + 1], boxes[i + 2])
            j = i + 1
            boxes[i], boxes[i + 1], boxes[i + 2] = boxes[i + o[0]], boxes[i + o[1]], boxes[i + o[2]]
            results[j], results[j + 1], results[j + 2] = results[j + o[0]], results[j + o[1]], results[j + o[2]]
        return results


def sort_boxes(raw_boxes, width, height, sorter, thres=0.8):
    """:516-534.  raw detector boxes ([x1, y1, x2, y2], any order) -> character boxes in reading order."""
    chars = clean_detections(raw_boxes, width, height)
    page = chars_to_columns([[list(b[0]), list(b[1])] for b in chars], width, height)
    ordered = []
    for _, col in sorter.predict(page).items():
        members = [_flat(b) for b in chars if box_iou(col, _flat(b), over_min=True) >= thres]
        ordered.extend(sorted(members, key=lambda b: (b[1] + b[3]) / 2))
    return ordered
