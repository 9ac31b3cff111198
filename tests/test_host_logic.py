"""Host logic either side of the hot path vs vectors produced by the reference's own functions
(scripts/make_golden_host.py): prompt assembly and page/character tiling geometry."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from callireader_amd import preprocess
from callireader_amd.conversation import get_conv_template

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'host_vectors.json'), encoding='utf-8'))


def test_prompt_strings():
    t = get_conv_template('internlm2-chat')
    t.append_message(t.roles[0], '<image>\n这幅书法作品内容是什么？' + '[UNUSED_TOKEN_140]' * 6)
    t.append_message(t.roles[1], None)
    assert t.get_prompt() == GOLD['prompt_single']
    t = get_conv_template('internlm2-chat')
    t.append_message(t.roles[0], 'q1'); t.append_message(t.roles[1], 'a1')
    t.append_message(t.roles[0], 'q2'); t.append_message(t.roles[1], None)
    assert t.get_prompt() == GOLD['prompt_history']
    assert t.sep == GOLD['sep'] == '<|im_end|>'


@pytest.mark.parametrize('size', sorted(GOLD['tiles']))
def test_dynamic_preprocess_tiles(size):
    w, h = map(int, size.split('x'))
    img = Image.fromarray((np.arange(h * w * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(h, w, 3))
    tiles = preprocess.dynamic_preprocess(img, image_size=448, use_thumbnail=True, max_num=12)
    exp = GOLD['tiles'][size]
    assert len(tiles) == exp['n_tiles']
    assert [hashlib.md5(np.asarray(t).tobytes()).hexdigest() for t in tiles] == exp['md5']


@pytest.mark.parametrize('size', sorted(GOLD.get('char_tiles', {})))
def test_load_image_2_geometry_pinned_to_the_reference(size):
    """The uint8 image the reference's load_image_2 (utils/utils.py:420-452) hands to its transform -- rescaled into
    [200, 350], centred on white, one 448x448 tile -- for crops in all three scale regimes and degenerate sizes."""
    w, h = (int(v) for v in size.split('x'))
    img = Image.fromarray((np.arange(h * w * 3, dtype=np.uint32) * 7 % 253).astype(np.uint8).reshape(h, w, 3))
    exp = GOLD['char_tiles'][size]
    padded = preprocess.pad_char(img)
    tiles = preprocess.dynamic_preprocess(padded, image_size=448, use_thumbnail=True, max_num=12)
    assert len(tiles) == exp['n_tiles'] == 1 and list(np.asarray(tiles[0]).shape[:2]) == exp['size']
    assert [hashlib.md5(np.asarray(t.convert('RGB')).tobytes()).hexdigest() for t in tiles] == exp['md5']
    # the planner of the GPU path states the same geometry
    job = preprocess.plan_char((0, 0, w, h), 0)
    nw, nh, left, top, _, _ = preprocess.char_canvas(w, h)
    assert (job['ow'], job['oh'], job['left'], job['top']) == (nw, nh, left, top)
    assert preprocess.load_image_2(img).shape == (1, 3, 448, 448)


def test_example_page_is_11_tiles_and_char_crop_is_one():
    assert preprocess.tile_grid(788, 2000) == (2, 5)                     # examples/0.jpg -> 10 + thumbnail
    assert preprocess.load_image(Image.new('RGB', (788, 2000))).shape == (11, 3, 448, 448)
    for wh in [(60, 90), (600, 300), (250, 260), (40, 1000)]:
        assert preprocess.load_image_2(Image.new('RGB', wh)).shape == (1, 3, 448, 448)


def test_transform_values():
    # white pixel -> (1 - mean) / std per channel; black -> -mean / std
    t = preprocess.build_transform(448)(Image.new('RGB', (10, 10), (255, 255, 255)))
    exp = (1 - torch.tensor(preprocess.IMAGENET_MEAN)) / torch.tensor(preprocess.IMAGENET_STD)
    assert torch.allclose(t[:, 0, 0], exp, atol=1e-6) and t.shape == (3, 448, 448)


def test_boxes_from_labelme():
    data = {'imageHeight': 2000, 'imageWidth': 788, 'shapes': [{'points': [[0.1, 0.2], [0.3, 0.4]]}, {'points': [[-0.1, 0.5], [1.2, 0.9]]}]}
    assert preprocess.boxes_from_labelme(data) == [(78, 400, 236, 800), (0, 1000, 788, 1800)]
