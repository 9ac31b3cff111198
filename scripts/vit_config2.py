#!/usr/bin/env python3
"""BASELINE config 2 alone (InternViT-300M encoder, 32 tiles 448x448, bf16): wall time per forward, and -- under
`rocprofv3 --kernel-trace --stats` -- the per-kernel breakdown of the same launches.
usage: python scripts/vit_config2.py [tiles=32] [iterations=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from callireader_amd.engine import Engine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dims = ModelDims.full()
eng = Engine(dims, max_pos=64)
for k, v in synthetic.iter_state_dict(dims, parts=('vit',), seed=0, device='cuda'):
    eng.load_weight(k, v)
eng.finalize()
px = synthetic.make_pixels(T, seed=0, device='cuda')
eng.vit_forward(px); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    eng.vit_forward(px)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f'ViT {T} tiles: {dt * 1e3:.2f} ms  {T / dt:.0f} tiles/s  {T * 723.6e9 / dt / 1e12:.0f} TFLOP/s = {T * 723.6e9 / dt / 2.5e15 * 100:.1f} % of 2.5 PFLOP/s'
      + (f'  [CR_VIT_CHUNK={os.environ["CR_VIT_CHUNK"]}]' if os.environ.get('CR_VIT_CHUNK') else ''))
