#!/usr/bin/env python3
"""Vendor yardsticks beside this engine's kernels, on the same box, same random data, interleaved rounds in one process
(VERDICT round 3, item 6).  Development aid: never imported by the package, never inside bench.py's timed region.

  GEMM:      torch.matmul (PyTorch-ROCm dispatches to hipBLASLt, then to rocBLAS with TORCH_BLAS_PREFER_HIPBLASLT=0 in a child) against
             cr_op_gemm with the plain-store epilogue, on the six dominant shapes of a benchmark step.
  attention: torch.nn.functional.scaled_dot_product_attention (flash = AOTriton's kernels; memory-efficient = CK / AOTriton, whichever this
             build carries) against cr_op_attention on the ViT shape and on the causal GQA prefill shape.  aiter is not in this image.
Writes gpurun_out/yardsticks.json (copied to profiles/round4/ by hand)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from callireader_amd import engine as E

GEMMS = [(64575, 3072, 1024, 'ViT QKV'), (64575, 1024, 1024, 'ViT proj'), (64575, 4096, 1024, 'ViT fc1'), (64575, 1024, 4096, 'ViT fc2'),
         (25312, 28672, 4096, 'prefill w1|w3'), (25312, 4096, 14336, 'prefill w2'), (4096, 4096, 4096, '4096^3'), (8192, 8192, 8192, '8192^3')]
ROUNDS = 5
g = torch.Generator(device='cuda').manual_seed(0)

def timeit(fn, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n

def med(x): return sorted(x)[len(x) // 2]

out = {'device': torch.cuda.get_device_name(0), 'torch': torch.__version__, 'blas': str(torch.backends.cuda.preferred_blas_library()),
       'rounds': ROUNDS, 'gemm': [], 'attention': []}
for (M, N, K, what) in GEMMS:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = (torch.rand(N, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    Wt = W.t()
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    Co = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)      # both sides write into a preallocated output (op_gemm would otherwise zero-fill a new one per call)
    ours = lambda: E.op_gemm(0, A, W, out=Co)
    vend = lambda: torch.matmul(A, Wt, out=C)
    ours(); vend(); torch.cuda.synchronize()
    n = 10 if M * N * K < 2e12 else 4
    t = {'ours': [], 'vendor': []}
    for _ in range(ROUNDS):
        t['ours'].append(timeit(ours, n)); t['vendor'].append(timeit(vend, n))
    fl = 2.0 * M * N * K
    rec = {'what': what, 'M': M, 'N': N, 'K': K, 'ours_ms': med(t['ours']), 'vendor_ms': med(t['vendor']),
           'ours_tflops': fl / med(t['ours']) / 1e9, 'vendor_tflops': fl / med(t['vendor']) / 1e9, 'ours_min_ms': min(t['ours']), 'vendor_min_ms': min(t['vendor'])}
    rec['vendor_over_ours'] = rec['vendor_tflops'] / rec['ours_tflops']
    ref = (A[:64].float() @ W.float().t())
    got = E.op_gemm(0, A, W)[:64].float()
    rec['ours_rel_l2_vs_fp32'] = float((got - ref).norm() / ref.norm())
    out['gemm'].append(rec)
    print(f"{what:14s} M={M} N={N} K={K}: ours {rec['ours_ms']:.3f} ms {rec['ours_tflops']:.0f} TF | torch.matmul {rec['vendor_ms']:.3f} ms {rec['vendor_tflops']:.0f} TF | vendor/ours {rec['vendor_over_ours']:.3f}", flush=True)
    del A, W, C, Co

def sdpa_backends():
    from torch.nn.attention import SDPBackend
    return [('flash', SDPBackend.FLASH_ATTENTION), ('efficient', SDPBackend.EFFICIENT_ATTENTION)]

from torch.nn.attention import sdpa_kernel
# ViT: 63 tiles x 16 heads x 1025 x 64, q pre-scaled by 2^-3 on our side = scale 1/8 on theirs
Bn, S, H, D = 63, 1025, 16, 64
qkv = torch.randn(Bn, S, 3 * H * D, device='cuda', generator=g).bfloat16()
o = torch.zeros(Bn, S, H * D, device='cuda', dtype=torch.bfloat16)
C3, C1 = 3 * H * D, H * D
ours_vit = lambda: E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
q4 = qkv[:, :, :C1].view(Bn, S, H, D).transpose(1, 2)
k4 = qkv[:, :, C1:2 * C1].view(Bn, S, H, D).transpose(1, 2)
v4 = qkv[:, :, 2 * C1:].view(Bn, S, H, D).transpose(1, 2)
fl = 4.0 * Bn * H * S * S * D
rec = {'what': 'ViT 63x16x1025x64 non-causal', 'flops': fl}
t = [];
ours_vit(); torch.cuda.synchronize()
for _ in range(ROUNDS): t.append(timeit(ours_vit, 20))
rec['ours_ms'] = med(t); rec['ours_tflops'] = fl / med(t) / 1e9
for name, be in sdpa_backends():
    try:
        with sdpa_kernel(be):
            fn = lambda: F.scaled_dot_product_attention(q4, k4, v4, scale=0.125)
            fn(); torch.cuda.synchronize()
            t = [timeit(fn, 20) for _ in range(ROUNDS)]
        rec[name + '_ms'] = med(t); rec[name + '_tflops'] = fl / med(t) / 1e9
    except Exception as e:
        rec[name + '_error'] = str(e)[:200]
out['attention'].append(rec); print(rec, flush=True)

# LLM prefill: one page, 32 q heads / 8 kv heads x 128, S = 3164, causal
S, NH, NKV, HD = 3164, 32, 8, 128
q = torch.randn(S, NH * HD, device='cuda', generator=g).bfloat16()
k = torch.randn(NKV, S, HD, device='cuda', generator=g).bfloat16()
v = torch.randn(NKV, S, HD, device='cuda', generator=g).bfloat16()
o2 = torch.zeros(S, NH * HD, device='cuda', dtype=torch.bfloat16)
ours_llm = lambda: E.op_attention(q, k, v, o2, [0, NH * HD, HD, 0, HD, S * HD, 0, HD, S * HD, 0, NH * HD, HD], 1, NH, S, S, HD, kv_group=NH // NKV, causal=True, s_div=11.313708498984761)
fl = 2.0 * NH * S * S * HD
rec = {'what': 'LLM prefill 32q/8kv x 3164 x 128 causal (half the square counted)', 'flops': fl}
ours_llm(); torch.cuda.synchronize()
t = [timeit(ours_llm, 20) for _ in range(ROUNDS)]
rec['ours_ms'] = med(t); rec['ours_tflops'] = fl / med(t) / 1e9
q4 = q.view(S, NH, HD).transpose(0, 1).unsqueeze(0)
k4 = k.unsqueeze(0); v4 = v.unsqueeze(0)
for name, be in sdpa_backends():
    for gqa in ('enable_gqa', 'expanded'):
        try:
            with sdpa_kernel(be):
                if gqa == 'enable_gqa':
                    fn = lambda: F.scaled_dot_product_attention(q4, k4, v4, is_causal=True, enable_gqa=True)
                else:
                    ke, ve = k4.repeat_interleave(NH // NKV, dim=1), v4.repeat_interleave(NH // NKV, dim=1)
                    fn = lambda: F.scaled_dot_product_attention(q4, ke, ve, is_causal=True)
                fn(); torch.cuda.synchronize()
                t = [timeit(fn, 20) for _ in range(ROUNDS)]
            rec[f'{name}_{gqa}_ms'] = med(t); rec[f'{name}_{gqa}_tflops'] = fl / med(t) / 1e9
        except Exception as e:
            rec[f'{name}_{gqa}_error'] = str(e)[:200]
out['attention'].append(rec); print(rec, flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/yardsticks.json', 'w'), indent=1)
