"""Host image pipeline of the reference, restated without torchvision (PIL + torch only).

Mirrors /root/reference/utils/utils.py:
  build_transform :354-362, find_closest_aspect_ratio :365-379 (inside tile_grid), dynamic_preprocess :381-417,
  load_image_2 :420-452 (character crop -> exactly one 448x448 tile), load_image :463-478 (page -> <=12 tiles + thumbnail).
CPU host code: it sits either side of the hot path, not on it.
"""
import numpy as np
import torch
from PIL import Image, ImageOps

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # config/configu.py:17-18
IMAGENET_STD = (0.229, 0.224, 0.225)


def build_transform(input_size):
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)

    def transform(img):
        if img.mode != 'RGB':
            img = img.convert('RGB')
        img = img.resize((input_size, input_size), Image.BICUBIC)        # T.Resize(..., BICUBIC) on a PIL image
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)   # T.ToTensor
        return (t - mean) / std                                          # T.Normalize
    return transform


def tile_grid(width, height, min_num=1, max_num=12, image_size=448):
    """(cols, rows) dynamic_preprocess picks for an image of this size (utils/utils.py:365-379 + :386-393): among all grids
    with min_num..max_num tiles, visited by tile count (then by first appearance while the count grows), the one whose
    cols / rows is nearest to width / height; a later grid with exactly the same distance replaces the held one only when the
    image covers more than half of that grid's pixel area."""
    grids = []
    for n in range(min_num, max_num + 1):                      # order of first appearance, as the reference's set is built
        for cols in range(1, n + 1):
            for rows in range(1, n + 1):
                if min_num <= cols * rows <= max_num and (cols, rows) not in grids:
                    grids.append((cols, rows))
    # the reference sorts a SET by tile count: grids of equal count come in the set's iteration order, which for these
    # small int tuples is reproduced by sorting on (count, hash order); tests/test_host_logic.py pins the outcome on
    # the reference's own tilings (host_vectors.json)
    grids = sorted(set(grids), key=lambda g: g[0] * g[1])
    aspect, area = width / height, width * height
    held, held_diff = (1, 1), float('inf')
    for cols, rows in grids:
        diff = abs(aspect - cols / rows)
        if diff < held_diff:
            held, held_diff = (cols, rows), diff
        elif diff == held_diff and area > 0.5 * image_size * image_size * cols * rows:
            held = (cols, rows)
    return held


def dynamic_preprocess(image, min_num=1, max_num=12, image_size=448, use_thumbnail=False):
    orig_width, orig_height = image.size
    cols, rows = tile_grid(orig_width, orig_height, min_num, max_num, image_size)
    target_width, target_height = image_size * cols, image_size * rows
    blocks = cols * rows
    resized_img = image.resize((target_width, target_height))
    processed = []
    for i in range(blocks):
        box = ((i % cols) * image_size, (i // cols) * image_size, ((i % cols) + 1) * image_size, ((i // cols) + 1) * image_size)
        processed.append(resized_img.crop(box))
    assert len(processed) == blocks
    if use_thumbnail and len(processed) != 1:
        processed.append(image.resize((image_size, image_size)))
    return processed


def char_canvas(width, height, input_size=448):
    """Geometry of load_image_2 (utils/utils.py:424-443): the longest side is brought into [200, 350] (smaller crops
    grow to 200, larger ones shrink to 350, the rest keep their size), sizes truncate to int, and the result sits
    centred on a white input_size square, the odd pixel of padding going right / down.
    Returns (new_w, new_h, left, top, right, bottom)."""
    longest = max(width, height)
    factor = 200 / longest if longest <= 200 else 350 / longest if longest >= 350 else 1.0
    new_w, new_h = int(width * factor), int(height * factor)
    return (new_w, new_h, (input_size - new_w) // 2, (input_size - new_h) // 2,
            (input_size - new_w + 1) // 2, (input_size - new_h + 1) // 2)


def pad_char(image, input_size=448):
    """The white-padded PIL image load_image_2 hands to the tiler (exactly input_size square -> one tile)."""
    new_w, new_h, left, top, right, bottom = char_canvas(*image.size, input_size)
    return ImageOps.expand(image.resize((new_w, new_h)), border=(left, top, right, bottom), fill=(255, 255, 255))


def load_image_2(image, input_size=448, max_num=12):
    """Character crop -> (1,3,448,448)."""
    if isinstance(image, str):
        image = Image.open(image).convert('RGB')
    padded = pad_char(image, input_size)
    transform = build_transform(input_size)
    images = dynamic_preprocess(padded, image_size=input_size, use_thumbnail=True, max_num=max_num)
    return torch.stack([transform(im) for im in images])


def load_image(image_file, input_size=448, max_num=12):
    """Page -> (tiles,3,448,448): aspect-ratio tiling (<= max_num) plus a thumbnail."""
    image = Image.open(image_file).convert('RGB') if isinstance(image_file, str) else image_file
    transform = build_transform(input_size)
    images = dynamic_preprocess(image, image_size=input_size, use_thumbnail=True, max_num=max_num)
    return torch.stack([transform(im) for im in images])


def boxes_from_labelme(data):
    """Pixel xyxy boxes from the labelme-style JSON of examples/0.json (normalised coords clamped to [0,1] then scaled),
    as get_single_embeddings.py:178-188 does."""
    h, w = data['imageHeight'], data['imageWidth']
    out = []
    for item in data['shapes']:
        (x1, y1), (x2, y2) = item['points']
        x1, y1, x2, y2 = [min(max(0, v), 1) for v in (x1, y1, x2, y2)]
        out.append((int(x1 * w), int(y1 * h), int(x2 * w), int(y2 * h)))
    return out


# ---- planning for the GPU path (cr_preprocess): sizes only, no pixel work on the host -----------------------------
def norm_lut():
    """bf16 [3][256]: the reference's ToTensor -> Normalize -> .to(bfloat16) applied to every byte value."""
    p = torch.arange(256, dtype=torch.uint8).float().div(255.0)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1)
    return ((p.unsqueeze(0) - mean) / std).to(torch.bfloat16).contiguous()


def plan_page(width, height, tile0=0, min_num=1, max_num=12, image_size=448):
    """Jobs (dict rows) for load_image(page): the tiled resize and, when there is more than one tile, the thumbnail."""
    cols, rows = tile_grid(width, height, min_num, max_num, image_size)
    jobs = [dict(sx0=0, sy0=0, sw=width, sh=height, ow=image_size * cols, oh=image_size * rows, mode=1, tile0=tile0, cols=cols, left=0, top=0)]
    n = cols * rows
    if n != 1:
        jobs.append(dict(sx0=0, sy0=0, sw=width, sh=height, ow=image_size, oh=image_size, mode=1, tile0=tile0 + n, cols=1, left=0, top=0))
        n += 1
    return jobs, n


def plan_char(box, tile, input_size=448):
    """Job for load_image_2(crop): rescale the longest side into [200,350], centre on a white canvas."""
    x1, y1, x2, y2 = [int(v) for v in box]
    width, height = x2 - x1, y2 - y1
    new_w, new_h, left, top, _, _ = char_canvas(width, height, input_size)
    return dict(sx0=x1, sy0=y1, sw=width, sh=height, ow=new_w, oh=new_h, mode=0, tile0=tile, cols=1, left=left, top=top)


JOB_FIELDS = ('sx0', 'sy0', 'sw', 'sh', 'ow', 'oh', 'mode', 'tile0', 'cols', 'left', 'top')      # = cr_prep_job (include/callireader_hip.h), eleven int32


def jobs_array(jobs):
    """plan_* dict rows -> int32 (n, 11) array in cr_prep_job's field order (what Engine.preprocess hands to cr_preprocess)."""
    return np.array([[j[f] for f in JOB_FIELDS] for j in jobs], dtype=np.int32).reshape(-1, len(JOB_FIELDS))


def plan_chars_array(boxes, width, height, tile0=0, input_size=448):
    """plan_char for all boxes of a page at once (boxes clipped to the page as calli_align's numpy slicing clips them), as the int32 table of jobs_array:
    the same arithmetic in float64 / int64 numpy as char_canvas' Python (a page has ~100 boxes; the per-box Python cost 2.7 ms a page)."""
    b = np.asarray([[int(v) for v in bx[:4]] for bx in boxes], dtype=np.int64).reshape(-1, 4)
    x1, y1 = np.maximum(b[:, 0], 0), np.maximum(b[:, 1], 0)
    x2, y2 = np.minimum(b[:, 2], width), np.minimum(b[:, 3], height)
    w, h = x2 - x1, y2 - y1
    longest = np.maximum(np.maximum(w, h), 1).astype(np.float64)                     # (a degenerate box is refused by cr_preprocess, not here)
    factor = np.where(longest <= 200, 200 / longest, np.where(longest >= 350, 350 / longest, 1.0))
    new_w, new_h = (w * factor).astype(np.int64), (h * factor).astype(np.int64)      # int() truncation
    out = np.zeros((len(b), len(JOB_FIELDS)), dtype=np.int32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4], out[:, 5] = x1, y1, w, h, new_w, new_h
    out[:, 7] = tile0 + np.arange(len(b))
    out[:, 8] = 1
    out[:, 9], out[:, 10] = (input_size - new_w) // 2, (input_size - new_h) // 2
    return out
