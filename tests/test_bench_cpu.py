"""bench.py's host-side pieces (no GPU): the script's flags, the benchlib modules, and the stand-in tokenizer directory of the api_level block."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_lists_every_block_switch_and_benchlib_imports():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    for flag in ('--gpus', '--steps', '--warmup', '--pages', '--no-api', '--api-batches', '--no-ragged', '--two-steps-one-decode', '--no-strong-share', '--no-traffic',
                 '--no-cpu-baseline', '--no-vit-extra', '--no-pipeline', '--scaling', '--total-pages'):
        assert flag in out, flag
    sys.path.insert(0, ROOT)
    import bench
    from benchlib import plan, measure, extras, api, kernels                      # noqa: F401  (every piece imports without a GPU)
    assert bench.plan_workload is plan.plan_workload and bench.CHAR_TILES == plan.CHAR_TILES == 96 and bench.PAGE_TILES == 11
    assert measure.PEAK_BF16_TFLOPS == 2500.0 and measure.PEAK_HBM_GBS == 8000.0
    for fn in ('strong_share_block', 'strong_scaling_block', 'single_gpu_extras', 'ragged_balanced', 'measure_balanced'):
        assert callable(getattr(extras, fn))


def test_api_level_tokenizer_dir_has_the_references_structure(tmp_path):
    """The stand-in for the reference's tokenizer files (benchlib/api.py): 92 544 sentencepiece pieces with [UNUSED_TOKEN_140] = 92 537 as a user-defined piece, the reference's
    added-token ids, every id of the 92 553-row vocabulary decodable (random-init weights pick any of them), and a page prompt with the reference's token counts."""
    pytest.importorskip('sentencepiece')
    sys.path.insert(0, ROOT)
    from benchlib.api import make_tokenizer_dir, make_pages, PROMPT, ADDED, VOCAB_SP
    from callireader_amd.tokenization_internlm2 import InternLM2Tokenizer
    from callireader_amd.conversation import get_conv_template
    from callireader_amd.config import IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, EOS_TOKEN_ID
    d = str(tmp_path / 'InternVL')
    os.makedirs(d)
    make_tokenizer_dir(d)
    tok = InternLM2Tokenizer.from_pretrained(d)
    assert tok.vocab_size == VOCAB_SP == 92544
    assert tok.convert_tokens_to_ids('[UNUSED_TOKEN_140]') == ALIGNED_TOKEN_ID == 92537
    assert tok.convert_tokens_to_ids('<IMG_CONTEXT>') == IMG_CONTEXT_TOKEN_ID == ADDED['<IMG_CONTEXT>'] and tok.convert_tokens_to_ids('<|im_end|>') == EOS_TOKEN_ID
    t = get_conv_template('internlm2-chat')
    t.append_message(t.roles[0], '<image>\n' + PROMPT + '[UNUSED_TOKEN_140]' * 288)
    t.append_message(t.roles[1], None)
    ids = tok(t.get_prompt().replace('<image>', '<img>' + '<IMG_CONTEXT>' * 256 * 11 + '</img>', 1), return_tensors='pt')['input_ids'].reshape(-1)
    assert int((ids == IMG_CONTEXT_TOKEN_ID).sum()) == 2816 and int((ids == ALIGNED_TOKEN_ID).sum()) == 288 and 3150 <= ids.numel() <= 3292     # the bench's KV cache has room for 3164 + 128
    text = tok.batch_decode(torch.arange(0, 92553).reshape(1, -1), skip_special_tokens=True)[0]
    assert isinstance(text, str) and len(text) > 92553
    assert tok.batch_decode(torch.tensor([[5, EOS_TOKEN_ID, 7]]), skip_special_tokens=False)[0].count('<|im_end|>') == 1
    # the page folder: the example page and its boxes JSON n times (what inference.py reads without a detector)
    paths, boxes = make_pages(str(tmp_path), 3, ROOT)
    assert len(paths) == 3 and len(boxes) == 96 and all(os.path.exists(os.path.splitext(p)[0] + '.json') for p in paths)
    from callireader_amd.inference import get_image_paths, boxes_for
    assert get_image_paths(str(tmp_path)) == sorted(paths) and boxes_for(paths[0]) == boxes
    assert json.load(open(os.path.splitext(paths[0])[0] + '.json'))['imageWidth'] == 788
