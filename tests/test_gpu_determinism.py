"""Run-to-run determinism of the whole stages at FULL depth: the same input through the same launches, back to back, must give the
same bits.  The kernels keep LDS-DMA in flight across raw barriers, wait for it by hand, vote on fallbacks and sum partials in a fixed
order -- a race in any of that shows as a rare difference between two runs (the ViT attention had one, once in a few launches:
tests/test_gpu_ops.py::test_attention_vit_launches_back_to_back_are_identical), which a comparison with a reference at a tolerance
does not see."""
import pytest
import torch

from callireader_amd.config import ModelDims, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model():
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    dims = ModelDims.full()
    m = InternVLChatModel.from_synthetic(dims, seed=0, device=0, max_tokens=3164 + 64, max_pages=2)
    m.img_context_token_id = IMG_CONTEXT_TOKEN_ID
    yield m


def test_vit_stage_is_deterministic(model):
    px = synthetic.make_pixels(96, seed=3, device=torch.device('cuda', 0))          # 64 + 32 tiles: a full chunk and a ragged one
    outs = [model.extract_feature(px).clone() for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_character_stage_is_deterministic(model):
    px = synthetic.make_pixels(96, seed=4, device=torch.device('cuda', 0))
    outs = []
    for _ in range(4):
        r, idx = model.align_tiles(px)
        outs.append((r.clone(), None if idx is None else idx.clone()))
    torch.cuda.synchronize()
    for r, idx in outs[1:]:
        assert torch.equal(r, outs[0][0])
        if idx is not None:
            assert torch.equal(idx, outs[0][1])


def test_prefill_and_decode_are_deterministic(model):
    import bench
    dev = torch.device('cuda', 0)
    dims = model.dims if hasattr(model, 'dims') else ModelDims.full()
    g = torch.Generator(device='cuda').manual_seed(7)
    ids = bench.build_ids(bench.PAGE_TILES, bench.CHAR_TILES, bench.TEXT_TOKENS, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, 1000).to(dev)
    vit = (torch.randn(bench.PAGE_TILES, 256, 4096, device=dev, generator=g) * 0.02).bfloat16()
    ref = (torch.randn(bench.CHAR_TILES, 3, 4096, device=dev, generator=g) * 0.02).bfloat16()
    e = model.engine.embed_splice(ids, vit, ref, img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID)
    runs = [model.generate_pages([e, e], max_new_tokens=24, eos_token_id=None) for _ in range(3)]
    for r in runs:
        assert r[0] == r[1]                                   # two copies of one page in one batch
        assert r == runs[0]
