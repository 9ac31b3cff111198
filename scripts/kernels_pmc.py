#!/usr/bin/env python3
"""The round's three headline kernels under `rocprofv3 --pmc` (development aid): ViT attention (63 tiles), the tiled GEMM on the ViT's
fc1 shape in bf16 and in e4m3 x e4m3, and on the LLM's w1|w3 shape likewise.
usage: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
       SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d <dir> -- python3 scripts/kernels_pmc.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
Bn, S, H, D = 63, 1025, 16, 64
qkv = (torch.randn(Bn, S, 3 * H * D, device='cuda', generator=g)).bfloat16()
o = torch.zeros(Bn, S, H * D, device='cuda', dtype=torch.bfloat16)
C3, C1 = 3 * H * D, H * D
for _ in range(2):
    E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
for (M, N, K, epi) in [(64575, 4096, 1024, 1), (50624, 28672, 4096, 4)]:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
    bias = None if epi == 4 else (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    a8, a_s = E.op_quantize_fp8(A)
    w8, w_s = E.op_quantize_fp8(W)
    for _ in range(2):
        E.op_gemm(epi, A, W, bias=bias)
        E.op_gemm_fp8x8(epi, a8, a_s, w8, w_s, bias=bias)
    del A, W, a8, w8
torch.cuda.synchronize()
