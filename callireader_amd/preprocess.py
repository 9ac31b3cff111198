"""Host image pipeline of the reference, restated without torchvision (PIL + torch only).

Mirrors /root/reference/utils/utils.py:
  build_transform :354-362, find_closest_aspect_ratio :365-379, dynamic_preprocess :381-417,
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


def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    best_ratio_diff = float('inf')
    best_ratio = (1, 1)
    area = width * height
    for ratio in target_ratios:
        target_aspect_ratio = ratio[0] / ratio[1]
        ratio_diff = abs(aspect_ratio - target_aspect_ratio)
        if ratio_diff < best_ratio_diff:
            best_ratio_diff = ratio_diff
            best_ratio = ratio
        elif ratio_diff == best_ratio_diff:
            if area > 0.5 * image_size * image_size * ratio[0] * ratio[1]:
                best_ratio = ratio
    return best_ratio


def tile_grid(width, height, min_num=1, max_num=12, image_size=448):
    """(cols, rows) chosen by dynamic_preprocess for an image of this size."""
    target_ratios = set((i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                        if min_num <= i * j <= max_num)
    target_ratios = sorted(target_ratios, key=lambda x: x[0] * x[1])
    return find_closest_aspect_ratio(width / height, target_ratios, width, height, image_size)


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


def load_image_2(image, input_size=448, max_num=12):
    """Character crop -> (1,3,448,448): rescale the longest side into [200,350], centre on white, one tile."""
    if isinstance(image, str):
        image = Image.open(image).convert('RGB')
    width, height = image.size
    if max(width, height) <= 200:
        scale = 200 / max(width, height)
    elif max(width, height) >= 350:
        scale = 350 / max(width, height)
    else:
        scale = 1.0
    new_w, new_h = int(width * scale), int(height * scale)
    image = image.resize((new_w, new_h))
    padded = ImageOps.expand(image, border=((input_size - new_w) // 2, (input_size - new_h) // 2,
                                            (input_size - new_w + 1) // 2, (input_size - new_h + 1) // 2),
                             fill=(255, 255, 255))
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
    if max(width, height) <= 200:
        scale = 200 / max(width, height)
    elif max(width, height) >= 350:
        scale = 350 / max(width, height)
    else:
        scale = 1.0
    new_w, new_h = int(width * scale), int(height * scale)
    return dict(sx0=x1, sy0=y1, sw=width, sh=height, ow=new_w, oh=new_h, mode=0, tile0=tile, cols=1,
                left=(input_size - new_w) // 2, top=(input_size - new_h) // 2)
