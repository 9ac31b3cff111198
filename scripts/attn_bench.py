#!/usr/bin/env python3
"""Attention microbenchmark (development aid): the ViT shape (63 tiles x 16 heads x 1025 x 64) and the LLM prefill shape."""
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
Bn, S, H, D = 63, 1025, 16, 64
qkv = (torch.randn(Bn, S, 3 * H * D, device='cuda', generator=g)).bfloat16()
o = torch.zeros(Bn, S, H * D, device='cuda', dtype=torch.bfloat16)
C3, C1 = 3 * H * D, H * D
run_vit = lambda: E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
# interleaved rounds in ONE process: attention_vit.hip (CLS peeled off, 8 x 16 full tiles) against the generic kernel (9 x 17)
res = {'vit': [], 'vit 2 waves': [], 'generic': []}
for rnd in range(5):
    for name, flag, nw in (('vit', '1', '4'), ('vit 2 waves', '1', '2'), ('generic', '0', '4')):
        os.environ['CR_VIT_ATTN'] = flag
        os.environ['CR_VIT_ATTN_NW'] = nw
        res[name].append(timeit(run_vit))
os.environ['CR_VIT_ATTN'] = '1'
os.environ['CR_VIT_ATTN_NW'] = '4'
o_new = o.clone(); run_vit(); torch.cuda.synchronize(); o_new = o.clone()
os.environ['CR_VIT_ATTN'] = '0'
run_vit(); torch.cuda.synchronize()
d = (o_new.float() - o.float()).abs()
os.environ['CR_VIT_ATTN'] = '1'
for name in res:
    ms = sorted(res[name])[len(res[name]) // 2]
    print(f'ViT attention 63x16x1025x64 [{name:11s}]: median {ms:.3f} ms (min {min(res[name]):.3f})  {4.0 * Bn * H * S * S * D / ms / 1e9:.1f} TFLOP/s')
print(f'  outputs of the two kernels: max |d| {float(d.max()):.4g}, mean |d| {float(d.mean()):.3g}, CLS rows max |d| {float(d[:, 0].max()):.4g}')
# LLM prefill: one page, 32 q heads / 8 kv heads x 128, S = 3164, causal
S, NH, NKV, HD = 3164, 32, 8, 128
q = torch.randn(S, NH * HD, device='cuda', generator=g).bfloat16()
k = torch.randn(NKV, S, HD, device='cuda', generator=g).bfloat16()
v = torch.randn(NKV, S, HD, device='cuda', generator=g).bfloat16()
o2 = torch.zeros(S, NH * HD, device='cuda', dtype=torch.bfloat16)
ms = timeit(lambda: E.op_attention(q, k, v, o2, [0, NH * HD, HD, 0, HD, S * HD, 0, HD, S * HD, 0, NH * HD, HD], 1, NH, S, S, HD, kv_group=NH // NKV, causal=True, s_div=11.313708498984761))
print(f'LLM causal prefill 32x3164x128: {ms:.3f} ms  {2.0 * NH * S * S * HD / ms / 1e9:.1f} TFLOP/s (causal half counted)')
