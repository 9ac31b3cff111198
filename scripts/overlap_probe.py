#!/usr/bin/env python3
"""Feasibility probe (development aid): batched decode on a CU-masked stream beside the ViT on the remaining CUs.
usage: CR_CUS=<cus for the ViT> python scripts/overlap_probe.py <decode_cus> [spread|block]"""
import sys, os, time, threading, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from callireader_amd.engine import Engine

DEC = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = sys.argv[2] if len(sys.argv) > 2 else 'spread'
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(on):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if on[w * 32 + b]) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


if MODE == 'prio':
    s_dec, s_vit = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
    dec_on = [True] * 256
    vit_on = [True] * 256
elif MODE == 'spread':
    step = 256 // DEC
    dec_on = [(i % step) == 0 for i in range(256)]
else:
    dec_on = [i < DEC for i in range(256)]
if MODE != 'prio':
    vit_on = [not x for x in dec_on]
    s_dec, s_vit = masked_stream(dec_on), masked_stream(vit_on)
dims = ModelDims.full()
ev = Engine(dims)
for k, v in synthetic.iter_state_dict(dims, parts=('vit', 'mlp1'), seed=0, device='cuda'):
    ev.load_weight(k, v)
ev.finalize()
el = Engine(dims, max_pos=4096)
for k, v in synthetic.iter_state_dict(dims, parts=('llm',), seed=0, device='cuda'):
    el.load_weight(k, v)
el.load_rope()
el.finalize()
px = synthetic.make_pixels(63, seed=0, device='cuda')
kv = el.kv_alloc(64, 3400)
emb = (torch.randn(3164, 4096, device='cuda') * 0.02).bfloat16()
for i0 in range(0, 64, 16):
    el.prefill_batch(kv, list(range(i0, i0 + 16)), [emb] * 16)
torch.cuda.synchronize()
live = list(range(64))


def vit_loop(n, stream, out):
    with torch.cuda.stream(stream):
        ev.vit_forward(px)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ev.vit_forward(px)
        stream.synchronize()
        out['vit_ms'] = (time.perf_counter() - t0) / n * 1e3


def dec_loop(n, stream, out):
    with torch.cuda.stream(stream):
        el.decode(kv, live)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            el.decode(kv, live)
        stream.synchronize()
        out['dec_ms'] = (time.perf_counter() - t0) / n * 1e3


r = {}
vit_loop(4, torch.cuda.current_stream(), r); print('ViT 63 tiles alone, default stream: %.2f ms' % r['vit_ms'], flush=True)
dec_loop(16, torch.cuda.current_stream(), r); print('decode step alone, default stream: %.2f ms' % r['dec_ms'], flush=True)
vit_loop(4, s_vit, r); print('ViT alone on its masked stream (%d CUs): %.2f ms' % (sum(vit_on), r['vit_ms']), flush=True)
dec_loop(16, s_dec, r); print('decode alone on its masked stream (%d CUs, %s): %.2f ms' % (DEC, MODE, r['dec_ms']), flush=True)
from callireader_amd import _binding as B
if os.environ.get('PROBE_PROF'):
    B.check(B.lib.cr_profile(ev._h, 1)); B.check(B.lib.cr_profile(el._h, 1))
if os.environ.get('PROBE_SINGLE'):
    # one host thread feeding both streams, the way bench.py's pipelined step does: 2 chunks, then 8 decode steps
    def single(n_slices):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nd = 0
        for _ in range(n_slices):
            with torch.cuda.stream(s_vit):
                for _ in range(int(os.environ.get('PROBE_CHUNKS', '2'))):
                    ev.vit_forward(px)
            with torch.cuda.stream(s_dec):
                for _ in range(int(os.environ.get('PROBE_STEPS', '4'))):
                    el.decode(kv, live); nd += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('single thread: %d chunks + %d decode steps in %.1f ms (alone they would take %.1f ms)' % (int(os.environ.get('PROBE_CHUNKS', '2')) * n_slices, nd, dt * 1e3, int(os.environ.get('PROBE_CHUNKS', '2')) * n_slices * r['vit_ms'] + nd * r['dec_ms']), flush=True)
    r['vit_ms'], r['dec_ms'] = 49.7, 10.76
    single(2); single(int(os.environ.get('PROBE_SLICES', '8')))
    sys.exit(0)
r2 = {}
NV, ND = int(os.environ.get('PROBE_NV', '8')), int(os.environ.get('PROBE_ND', '24'))
ta = threading.Thread(target=vit_loop, args=(NV, s_vit, r2))
tb = threading.Thread(target=dec_loop, args=(ND, s_dec, r2))
ta.start(); tb.start(); ta.join(); tb.join()
print('together: ViT %.2f ms per 63 tiles, decode %.2f ms per step' % (r2['vit_ms'], r2['dec_ms']), flush=True)
