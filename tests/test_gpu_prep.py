"""GPU tile preprocessing (cr_preprocess) vs the reference's host pipeline (PIL resize + ToTensor/Normalize + bf16):
integer/byte work end to end, so the bar is BIT-EXACT equality of the bf16 tiles."""
import numpy as np
import pytest
import torch
from PIL import Image

from callireader_amd import preprocess
from callireader_amd.config import ModelDims

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from callireader_amd.engine import Engine
    return Engine(ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1))


def rand_image(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
    img = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w]            # blocky structure + noise: edges and ramps
    return (img.astype(np.int32) + rng.integers(-20, 21, (h, w, 3))).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize('w,h', [(788, 2000), (640, 500), (448, 448), (1000, 1000), (3000, 500), (300, 260)])
def test_page_tiles_bit_exact(eng, w, h):
    arr = rand_image(w, h, w + h)
    ref = preprocess.load_image(Image.fromarray(arr)).to(torch.bfloat16)
    jobs, n = preprocess.plan_page(w, h)
    got = eng.preprocess(torch.from_numpy(arr), jobs, n)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert torch.equal(got.cpu(), ref)


def test_char_tiles_bit_exact_all_scale_regimes(eng):
    arr = rand_image(900, 1400, 7)
    boxes = [(10, 20, 110, 140),        # <= 200: upscale
             (200, 50, 420, 300),       # 200..350: untouched
             (300, 310, 380, 1300),     # >= 350: downscale, tall and thin
             (0, 0, 900, 1400),         # whole page as one crop
             (5, 5, 6, 205),            # 1 pixel wide -> width int(1*1.0) = 1
             (100, 100, 449, 449),      # 349: just under the upper threshold
             (100, 100, 450, 450)]      # 350: exactly at it
    jobs = [preprocess.plan_char(b, i) for i, b in enumerate(boxes)]
    got = eng.preprocess(torch.from_numpy(arr), jobs, len(boxes))
    torch.cuda.synchronize()
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        ref = preprocess.load_image_2(Image.fromarray(arr[y1:y2, x1:x2])).to(torch.bfloat16)[0]
        assert torch.equal(got[i].cpu(), ref), (i, boxes[i])


def test_page_and_chars_in_one_call_and_errors(eng):
    from callireader_amd._binding import CalliReaderError
    arr = rand_image(640, 500, 3)
    jobs, n = preprocess.plan_page(640, 500)
    boxes = [(10, 20, 110, 140), (200, 50, 420, 300)]
    jobs += [preprocess.plan_char(b, n + i) for i, b in enumerate(boxes)]
    got = eng.preprocess(torch.from_numpy(arr), jobs, n + len(boxes))
    torch.cuda.synchronize()
    assert torch.equal(got[:n].cpu(), preprocess.load_image(Image.fromarray(arr)).to(torch.bfloat16))
    assert torch.equal(got[n].cpu(), preprocess.load_image_2(Image.fromarray(arr[20:140, 10:110])).to(torch.bfloat16)[0])
    with pytest.raises(CalliReaderError):
        eng.preprocess(torch.from_numpy(arr), [preprocess.plan_char((600, 400, 700, 520), 0)], 1)      # box outside the page
    with pytest.raises(CalliReaderError):
        eng.preprocess(torch.from_numpy(arr), jobs, 2)                                                 # not enough output tiles
