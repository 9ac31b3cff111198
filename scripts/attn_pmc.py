#!/usr/bin/env python3
"""ViT attention under `rocprofv3 --pmc` (development aid): three launches of the 63-tile shape.
usage: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
       SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d <dir> -- python3 scripts/attn_pmc.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
Bn, S, H, D = 63, 1025, 16, 64
qkv = (torch.randn(Bn, S, 3 * H * D, device='cuda', generator=g)).bfloat16()
o = torch.zeros(Bn, S, H * D, device='cuda', dtype=torch.bfloat16)
C3, C1 = 3 * H * D, H * D
for _ in range(3):
    E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
torch.cuda.synchronize()
