#!/usr/bin/env python3
"""The resampler on 252 tiles (one RS_CHUNK) with the current perceiver attention kernel and with round 1's (CR_PERCEIVER_ATTN_V1=1): wall time of cr_resample;
under `rocprofv3 --kernel-trace --stats` the per-kernel view."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd.engine import Engine
from callireader_amd import synthetic
dims = ModelDims.full()
eng = Engine(dims)
for k, v in synthetic.iter_state_dict(dims, parts=('resampler',), seed=0, device=eng.device):
    eng.load_weight(k, v)
eng.finalize()
os.environ['CR_PERCEIVER_ATTN_V1'] = '1'
old = Engine(dims)
old.share_weights_from(eng)
feats = (torch.randn(252, 256, 4096, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0)) * 0.7).bfloat16()
for name, e in (('current', eng), ('round 1', old)):
    e.resample(feats); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): e.resample(feats)
    torch.cuda.synchronize()
    print(f'{name}: resample of 252 tiles {1e3 * (time.perf_counter() - t0) / 5:.3f} ms (4 layers: 4 perceiver attention launches inside)')
print('bits equal:', torch.equal(eng.resample(feats), old.resample(feats)))
