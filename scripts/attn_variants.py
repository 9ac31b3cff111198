#!/usr/bin/env python3
"""Development aid: A/B of the ViT attention launch under CR_VIT_ATTN values (1 = attention_vit.hip, 0 = the generic kernel; experiment builds add more) in one process, interleaved rounds, 63 and 255 tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n
for Bn in (63, 255):
    S, H, D = 1025, 16, 64
    qkv = (torch.randn(Bn, S, 3 * H * D, device='cuda', generator=g)).bfloat16()
    o = torch.zeros(Bn, S, H * D, device='cuda', dtype=torch.bfloat16)
    C3, C1 = 3 * H * D, H * D
    run = lambda: E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
    names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['1', '0']
    res = {n: [] for n in names}
    outs = {}
    for rnd in range(5):
        for n in names:
            os.environ['CR_VIT_ATTN'] = n
            res[n].append(timeit(run))
            if rnd == 0:
                torch.cuda.synchronize(); outs[n] = o.clone()
    for n in names:
        ms = sorted(res[n])[2]
        d = float((outs[n].float() - outs[names[0]].float()).abs().max())
        print(f'{Bn} tiles  variant {n}: median {ms:.4f} ms (min {min(res[n]):.4f})  {4.0 * Bn * H * S * S * D / ms / 1e9:.1f} TFLOP/s   max|d| vs first {d:.3g}')
os.environ['CR_VIT_ATTN'] = '1'
