#!/usr/bin/env python3
"""LLM causal prefill attention under rocprofv3 --pmc (development aid): 16 pages x 32 heads x 3164 x 128, one launch shape of the bench."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
S, NH, NKV, HD = 3164, 32, 8, 128
q = torch.randn(S, NH * HD, device='cuda', generator=g).bfloat16()
k = torch.randn(NKV, S, HD, device='cuda', generator=g).bfloat16()
v = torch.randn(NKV, S, HD, device='cuda', generator=g).bfloat16()
o2 = torch.zeros(S, NH * HD, device='cuda', dtype=torch.bfloat16)
for _ in range(3):
    E.op_attention(q, k, v, o2, [0, NH * HD, HD, 0, HD, S * HD, 0, HD, S * HD, 0, NH * HD, HD], 1, NH, S, S, HD, kv_group=NH // NKV, causal=True, s_div=11.313708498984761)
torch.cuda.synchronize()
