#!/usr/bin/env python3
"""Run under torch.distributed.run: parallel.chat_ocr_pages_sharded (detector -> reading order -> sharded character tiles -> one all-gather -> page owners) must give
every page the response of the single-process model.chat_ocr_pages.  CR_CKPT_DIR / CR_PARAMS_DIR: a checkpoint on disk in the reference's layout
(tests/test_gpu_boundary.py writes a synthetic one).  Backend: CR_DIST_BACKEND (gloo lets the ranks share GPU 0 on a single-GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from PIL import Image

from callireader_amd.modeling_internvl_chat import InternVLChatModel
from callireader_amd.tokenization_internlm2 import InternLM2Tokenizer
from callireader_amd.parallel import chat_ocr_pages_sharded, plan_balanced, plan_even
from callireader_amd.preprocess import plan_page

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
dist.init_process_group(os.environ.get('CR_DIST_BACKEND', 'nccl'))


class _Box:
    def __init__(self, b):
        self.xyxy = torch.tensor([b], dtype=torch.float32)


class _Res:
    def __init__(self, boxes):
        self.boxes = [_Box(b) for b in boxes]


class Detector:
    """ultralytics.YOLO's call shape; the boxes depend on the page (its width), in no particular order."""
    def __call__(self, arr, verbose=False):
        w = arr.shape[1]
        raw = [[300, 310, 380, 480], [10, 20, 110, 140], [200, 50, 420, 300], [12, 160, 108, 300]]
        if w > 700:
            raw = raw[:3] + [[500, 40, 640, 200], [520, 230, 650, 420]]
        return [_Res(raw)]


ckpt, params = os.environ['CR_CKPT_DIR'], os.environ['CR_PARAMS_DIR']
m = InternVLChatModel.from_pretrained(ckpt, params_dir=params, torch_dtype=torch.bfloat16, max_tokens=4096, max_pages=4).eval().cuda()
tok = InternLM2Tokenizer.from_pretrained(ckpt)
m.aligned_token_id = tok.convert_tokens_to_ids('[UNUSED_TOKEN_140]')
rng = np.random.default_rng(7)
images = [Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)) for h, w in ((500, 640), (460, 900), (520, 660))]
gen = dict(num_beams=1, max_new_tokens=5, do_sample=False)
q = '读出图中所有文字。'
det = Detector()
outs = [chat_ocr_pages_sharded(m, tok, det, images, q, gen, repetition_penalty=1.0)]
# fixed plans need the pages' sizes: boxes from a first pass are not exposed, so re-derive the sizes the way the function does
n_chars = [4, 5, 4]
tiles = [plan_page(*im.size)[1] for im in images]
toks = [60 + 256 * t + 3 * c for t, c in zip(tiles, n_chars)]           # (only the plan's cost estimate reads these)
for pl in (plan_balanced(3, world, tiles, n_chars, toks, 5, owners=1), plan_even(3, world, tiles, n_chars, toks, 5)):
    outs.append(chat_ocr_pages_sharded(m, tok, det, images, q, gen, repetition_penalty=1.0, plan=pl))
# round 6: the pages as FILES (sizes from the headers, pixels decoded on threads only where a rank needs them), boxes handed in, the plan under measured stage costs
import tempfile
tmp = tempfile.mkdtemp(prefix=f'cr_dist_chat_{rank}_')
paths = []
for k, im in enumerate(images):
    paths.append(os.path.join(tmp, f'p{k}.png'))
    im.save(paths[-1])
from callireader_amd import ordering
boxes = [[[int(v) for v in b[:4]] for b in ordering.acquire_boxes(det, im, m.sorter)] for im in images]
outs.append(chat_ocr_pages_sharded(m, tok, None, paths, q, gen, boxes_list=boxes, repetition_penalty=1.0, cost='measure'))


# a page nobody can read, and a detector that fails on ONE rank's page: every rank raises the same RuntimeError after the exchange (nobody waits in a collective)
class FailsOnWide(Detector):
    def __call__(self, arr, verbose=False):
        if arr.shape[1] > 700:
            raise ValueError('detector down')
        return super().__call__(arr, verbose)


errs = []
for args, kw in (((None, paths + [os.path.join(tmp, 'missing.png')]), dict(boxes_list=boxes + [boxes[0]])), ((FailsOnWide(), images), {})):
    try:
        chat_ocr_pages_sharded(m, tok, args[0], args[1], q, gen, repetition_penalty=1.0, **kw)
        errs.append(None)
    except RuntimeError as e:
        errs.append(str(e))
all_errs = [None] * world
dist.all_gather_object(all_errs, errs)
same_errors = all(e == all_errs[0] for e in all_errs) and all(x is not None for x in errs) and 'cannot be read' in errs[0] and 'detection failed' in errs[1]
if rank == 0:
    single = m.chat_ocr_pages(tok, det, images, q, gen, repetition_penalty=1.0)
    ok = all(o == single for o in outs) and same_errors
    print('DIST_CHAT', 'OK' if ok else 'MISMATCH', outs, single, flush=True)
dist.barrier()
dist.destroy_process_group()
