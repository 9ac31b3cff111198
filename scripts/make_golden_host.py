#!/usr/bin/env python3
"""Golden vectors for the HOST logic either side of the hot path, produced by the reference's own functions
(build container only): prompt assembly (InternVL/conversation.py), tiling (utils/utils.py dynamic_preprocess) and the
character-crop geometry of load_image_2 (utils/utils.py:420-452: rescale into [200, 350], white padding, one tile).
torchvision is absent here; its three transforms are stood in for by their documented arithmetic (Resize to the size the
tile already has = identity, ToTensor = uint8 / 255 as CHW float32, Normalize = (x - mean) / std), so what is pinned for
load_image_2 is the uint8 image the reference hands to the transform (geometry + Pillow's resize) and the tile count."""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs  # noqa: E402

install_stubs()
import types  # noqa: E402
tvt = sys.modules['torchvision.transforms']
CAPTURED = []


def _compose(steps):
    def run(img):
        CAPTURED.append(np.asarray(img.convert('RGB')).copy())          # what reaches the transform
        import torch
        return torch.zeros(3, 448, 448)
    return run


tvt.Compose = _compose
tvt.Lambda = tvt.Resize = tvt.ToTensor = tvt.Normalize = lambda *a, **k: None
sys.modules['torchvision'].transforms = tvt
for name in ['sklearn', 'sklearn.cluster', 'tqdm']:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)

from InternVL.conversation import get_conv_template  # noqa: E402
from utils.utils import dynamic_preprocess, find_closest_aspect_ratio, load_image_2  # noqa: E402

out = {}
t = get_conv_template('internlm2-chat')
t.append_message(t.roles[0], '<image>\n这幅书法作品内容是什么？' + '[UNUSED_TOKEN_140]' * 6)
t.append_message(t.roles[1], None)
out['prompt_single'] = t.get_prompt()
t = get_conv_template('internlm2-chat')
t.append_message(t.roles[0], 'q1'); t.append_message(t.roles[1], 'a1')
t.append_message(t.roles[0], 'q2'); t.append_message(t.roles[1], None)
out['prompt_history'] = t.get_prompt()
out['sep'] = t.sep

grids = {}
for (w, h) in [(788, 2000), (448, 448), (1000, 1000), (3000, 500), (500, 3000), (1344, 896), (640, 480), (100, 900), (2000, 788)]:
    img = Image.fromarray((np.arange(h * w * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(h, w, 3))
    tiles = dynamic_preprocess(img, image_size=448, use_thumbnail=True, max_num=12)
    grids[f'{w}x{h}'] = {'n_tiles': len(tiles),
                         'md5': [hashlib.md5(np.asarray(ti).tobytes()).hexdigest() for ti in tiles]}
out['tiles'] = grids

chars = {}
for (w, h) in [(60, 90), (200, 200), (201, 120), (349, 349), (350, 100), (351, 700), (97, 33), (1, 1), (1000, 3), (448, 448), (123, 457)]:
    img = Image.fromarray((np.arange(h * w * 3, dtype=np.uint32) * 7 % 253).astype(np.uint8).reshape(h, w, 3))
    del CAPTURED[:]
    px = load_image_2(img)
    assert len(CAPTURED) == px.shape[0]
    chars[f'{w}x{h}'] = {'n_tiles': int(px.shape[0]), 'size': list(CAPTURED[0].shape[:2]),
                         'md5': [hashlib.md5(c.tobytes()).hexdigest() for c in CAPTURED]}
out['char_tiles'] = chars
json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'host_vectors.json'), 'w'), ensure_ascii=False, indent=1)
print('ok', {k: v['n_tiles'] for k, v in grids.items()})
