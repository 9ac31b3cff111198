#!/usr/bin/env python3
"""Quick timing of the ViT (config 2: 32 tiles, 24 layers) on one GPU; development aid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from callireader_amd.engine import Engine

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dims = ModelDims.full()
eng = Engine(dims)
for k, v in synthetic.iter_state_dict(dims, parts=('vit', 'mlp1'), seed=0, device='cuda'):
    eng.load_weight(k, v)
eng.finalize()
px = synthetic.make_pixels(T, seed=0, device='cuda')
out = eng.vit_forward(px)
torch.cuda.synchronize()
print('finite', bool(torch.isfinite(out.float()).all()), 'absmax', float(out.float().abs().max()))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
ev[0].record()
for i in range(iters):
    eng.vit_forward(px)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
best = min(ms)
fl = T * 723.6e9
print(f'T={T} ms/iter {ms}  best {best:.2f} ms  {T / best * 1e3:.1f} tiles/s  {fl / best / 1e9:.1f} TFLOP/s  frac of 2.5PF {fl / best / 1e9 / 2500:.3f}')
