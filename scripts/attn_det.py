#!/usr/bin/env python3
"""Development aid: is the ViT attention launch deterministic, and where does it differ from the generic kernel?
Launches go out back to back (no synchronisation in between) into separate output buffers, several rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
S, H, D = 1025, 16, 64
C3, C1 = 3 * H * D, H * D
bad = 0
for Bn in (63, 255, 32, 255):
    qkv = (torch.randn(Bn, S, 3 * H * D, device='cuda', generator=g)).bfloat16()
    def run(flag, o):
        os.environ['CR_VIT_ATTN'] = flag
        E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
    ref = torch.full((Bn, S, H * D), 7.0, device='cuda', dtype=torch.bfloat16)
    run('0', ref)
    torch.cuda.synchronize()
    for rnd in range(3):
        outs = [torch.full((Bn, S, H * D), 7.0, device='cuda', dtype=torch.bfloat16) for _ in range(8)]
        for o in outs: run('1', o)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            d = (o.float() - ref.float()).abs()
            ne = int((o != outs[0]).sum())
            if ne or float(d.max()) > 0.0079:
                bad += 1
                big = (d > 0.0079).nonzero()
                print(f'{Bn} tiles round {rnd} launch {i}: max|d| vs generic {float(d.max()):.4g}; differing from launch 0: {ne}; > 0.0079: {big.shape[0]}', big[:8].tolist(), flush=True)
    print(f'{Bn} tiles: done', flush=True)
print('BAD LAUNCHES', bad)
