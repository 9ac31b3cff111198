"""Host side of a BATCH of pages (new: the reference loads one page per chat_ocr call -- Image.open + the per-box PIL loop + one .cuda() per tile,
/root/reference/InternVL/modeling_internvl_chat.py:580-585,664-671 -- and inference.py:47-59 loops over it).

What a page costs on the host before any GPU work exists (examples/0.jpg, this container): JPEG decode 14.5 ms, job planning 2.7 ms, tokenising the
3 158-token prompt 10.6 ms: 28 ms per page, 1.8 s per 64-page batch when it is done page after page on the thread that also feeds the GPU.  And every pageable
host-to-device copy on the compute stream (the page's pixels, the job table of cr_preprocess, the prompt ids) parks that thread until the stream has drained.
Here:
  * `decode()`   image files -> RGB bytes in pinned memory on a thread pool (PIL's decoders release the GIL); started one batch AHEAD by chat_ocr_stream;
  * `tiles()`    per page ONE asynchronous upload + the tile kernels of cr_preprocess, on a stream and a context of their own (the context owns the workspace the
                 kernels use, so nothing here touches what the compute stream is using; the detector / OrderFormer, when boxes have to be found, run there too);
                 all tiles of the batch are written straight into two tensors (page tiles, character tiles: no torch.cat), and the compute stream waits for
                 ONE event;
  * job tables   as int32 arrays (numpy arithmetic = the per-box Python of preprocess.plan_char, pinned by tests/test_host_logic.py).
Per page the tiles are the bits `chat_ocr` makes for that page: the same kernels on the same bytes.  A page that cannot be read or has no box fails ALONE
(`errors='return'`): folder mode's per-image `try/except -> "ERROR!"` (inference.py:55-57) survives batching."""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image

from .engine import Engine
from .preprocess import plan_page, plan_chars_array, jobs_array


def _decode_one(im, device):
    """-> (PIL page, pinned uint8 (H, W, 3) tensor, seconds) or (None, exception, seconds)"""
    t0 = time.perf_counter()
    try:
        page = Image.open(im).convert('RGB') if isinstance(im, str) else im.convert('RGB')
        arr = torch.from_numpy(np.array(page))
        if device is not None:
            torch.cuda.set_device(device)
            arr = arr.pin_memory()
        return page, arr, time.perf_counter() - t0
    except Exception as e:                                            # chat_ocr turns any failure to load the page into FileNotFoundError (:670-671)
        err = FileNotFoundError(str(e) or repr(e))
        err.__cause__ = e
        return None, err, time.perf_counter() - t0


_POOL = None


def decode_pages(images, device, workers=None):
    """[(PIL page, pinned uint8 tensor, seconds) | (None, exception, seconds)] for a list of paths / PIL images, decoded on a module-level thread pool (callers without a
    PageFeeder: parallel.chat_ocr_pages_sharded)."""
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(workers or min(8, os.cpu_count() or 1), thread_name_prefix='cr-decode')
    return [f.result() for f in [_POOL.submit(_decode_one, im, device) for im in images]]


class PageTiles:
    """One batch after tiles(): `ok` = indices (into the batch) of the pages that made it, `failed` = {index: exception}; page_px / char_px hold the tiles of
    the ok pages in order (n_tiles / n_chars per ok page); `ready` = event on the feeder's stream after the last tile kernel."""
    __slots__ = ('ok', 'failed', 'page_px', 'char_px', 'n_tiles', 'n_chars', 'ready', 'sizes')


class PageFeeder:
    def __init__(self, model, workers=None):
        eng = model.engine
        self.m = model
        self.device = eng.device
        self.io = Engine(eng.dims, device=eng.device.index, max_pos=64)       # own workspace; borrows the OrderFormer weights when there is a sorter
        self.stream = torch.cuda.Stream(device=eng.device)
        self.pool = ThreadPoolExecutor(workers or min(8, os.cpu_count() or 1), thread_name_prefix='cr-decode')
        self._sorter_of, self._sorter, self._shared_version = None, None, -1
        self.stats = {'pages': 0, 'decode_s': 0.0, 'decode_wait_s': 0.0, 'detect_s': 0.0, 'plan_s': 0.0, 'enqueue_s': 0.0}

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        if self.io is not None:
            self.stream.synchronize()
            self.io.close()
            self.io = None

    def decode(self, images):
        """Start decoding a batch; returns the handle tiles() takes."""
        return [self.pool.submit(_decode_one, im, self.device) for im in images]

    def _io_sorter(self):
        """The model's OrderFormer bound to the feeder's context (same weights, own workspace), so that finding boxes does not wait for the compute stream."""
        src = self.m.sorter
        if src is None:
            return None
        ver = self.m.engine.weights_version
        if self._sorter_of is not src or self._shared_version != ver:      # (a reload / re-finalize / fp8 switch of the owner invalidates what a borrower holds)
            from .ordering import OrderFormer
            self.stream.synchronize()
            self.io.share_weights_from(self.m.engine)
            self._sorter, self._sorter_of, self._shared_version = OrderFormer(self.io, max_nums=src.max_nums), src, ver
        return self._sorter

    def tiles(self, decoded, boxes_list=None, detect_model=None, use_p=True, errors='raise'):
        from . import ordering
        st = self.stats
        t0 = time.perf_counter()
        loaded = [f.result() for f in decoded]
        st['decode_wait_s'] += time.perf_counter() - t0
        st['decode_s'] += sum(x[2] for x in loaded)
        st['pages'] += len(loaded)
        out = PageTiles()
        out.failed, plans = {}, []

        def fail(i, e):
            if errors == 'raise':
                raise e
            out.failed[i] = e

        with torch.cuda.stream(self.stream):
            for i, (page, arr, _) in enumerate(loaded):
                if page is None:
                    fail(i, arr)
                    continue
                w, h = page.size
                cj = None
                if use_p:
                    bx = boxes_list[i] if boxes_list is not None else None
                    try:
                        if bx is None:
                            t0 = time.perf_counter()
                            bx = ordering.acquire_boxes(detect_model, page, self._io_sorter())      # exactly what this page's own chat_ocr call does (:346-394, :558)
                            st['detect_s'] += time.perf_counter() - t0
                        if len(bx) == 0:
                            raise RuntimeError('calli_align: no character box on the page (the reference fails here too: torch.cat() of an empty list, '
                                               'modeling_internvl_chat.py:585)')
                        t0 = time.perf_counter()
                        cj = plan_chars_array(bx, w, h)
                        st['plan_s'] += time.perf_counter() - t0
                    except Exception as e:
                        fail(i, e)
                        continue
                t0 = time.perf_counter()
                pj, n = plan_page(w, h)
                plans.append((i, arr, jobs_array(pj), n, cj))
                st['plan_s'] += time.perf_counter() - t0
            t0 = time.perf_counter()
            size = self.m.dims.image_size
            tot_p, tot_c = sum(p[3] for p in plans), sum(len(p[4]) for p in plans if p[4] is not None)
            out.page_px = torch.empty(tot_p, 3, size, size, device=self.device, dtype=torch.bfloat16)
            out.char_px = torch.empty(tot_c, 3, size, size, device=self.device, dtype=torch.bfloat16) if use_p else None
            out.ok, out.n_tiles, out.n_chars, out.sizes = [], [], [], []
            po = co = 0
            for i, arr, pj, n, cj in plans:
                try:
                    dev = arr.to(self.device, non_blocking=True)
                    nc = len(cj) if cj is not None else 0
                    # (all of a page's rectangles are validated before anything is launched: a refused page leaves no tile half-written)
                    if nc:
                        cj[:, 7] += co                                # tile0: this page's slot in the batch's tensors
                        self.io.preprocess(dev, cj, tot_c, out=out.char_px)
                    pj[:, 7] += po
                    self.io.preprocess(dev, pj, tot_p, out=out.page_px)
                except Exception as e:
                    fail(i, e)
                    continue
                out.ok.append(i); out.n_tiles.append(n); out.n_chars.append(nc); out.sizes.append(loaded[i][0].size)
                po += n; co += nc
            # a page refused by cr_preprocess after the tensors were sized (a degenerate box) took no slot: the written part is the prefix
            out.page_px = out.page_px[:po]
            if out.char_px is not None:
                out.char_px = out.char_px[:co]
            out.ready = torch.cuda.Event()
            out.ready.record(self.stream)
            st['enqueue_s'] += time.perf_counter() - t0
        return out

    def hand_over(self, batch, stream=None):
        """Make `stream` (default: the current one) wait for the batch's tiles; the tensors were allocated on the feeder's stream."""
        stream = stream or torch.cuda.current_stream()
        stream.wait_event(batch.ready)
        for t in (batch.page_px, batch.char_px):
            if t is not None and t.numel():
                t.record_stream(stream)
