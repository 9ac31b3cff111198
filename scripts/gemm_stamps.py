#!/usr/bin/env python3
"""Development aid: in-kernel timeline of gemm256 (library built with -DCR_DIAG_STAMPS for gemm256.hip:
python scripts/build_variant.py gstamp gemm256.hip -DCR_DIAG_STAMPS=1; CR_HIP_LIB=ab/libgstamp.so python scripts/gemm_stamps.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64575, 4096, 1024)))
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 1
g = torch.Generator(device='cuda').manual_seed(0)
A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
bias = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
dbg = torch.zeros(256 * 2 * 32 * 4, device='cuda', dtype=torch.int64)
res = torch.rand(M, N, device='cuda', generator=g).bfloat16() if epi == 3 else None
for _ in range(20):
    E.op_gemm(epi, A, W, bias=bias, res=res)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
E.op_gemm(epi, A, W, bias=bias, scale=dbg, res=res)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1])
d = dbg.cpu().reshape(256, 2, 32, 4)
ok = d[..., 3] > 0
st, en = d[..., 0][ok].min().item(), d[..., 3][ok].max().item()
print(f'M={M} N={N} K={K} epi={epi}: {ms:.3f} ms by events; first stamp -> last stamp {en - st} ticks -> {(en - st) / ms / 1e6:.3f} GHz if ticks were core clocks')
ntile = ok[:, 0].sum(dim=1)
print('tiles per workgroup: min', int(ntile.min()), 'max', int(ntile.max()))
main = (d[..., 1] - d[..., 0])[ok].double(); drain = (d[..., 2] - d[..., 1])[ok].double(); epil = (d[..., 3] - d[..., 2])[ok].double()
print(f'per tile (ticks): main loop {main.mean():.0f} (min {main.min():.0f} max {main.max():.0f}), drain {drain.mean():.0f}, epilogue {epil.mean():.0f} (min {epil.min():.0f} max {epil.max():.0f})')
gaps = []
for b in range(256):
    n = int(ntile[b])
    for t in range(n - 1):
        gaps.append(int(d[b, 0, t + 1, 0] - d[b, 0, t, 3]))
gaps = torch.tensor(gaps).double()
print(f'gap between a tile\'s end and the next start: mean {gaps.mean():.0f}')
first = d[:, 0, 0, 0].double(); last = torch.stack([d[b, 0, int(ntile[b]) - 1, 3] for b in range(256)]).double()
print(f'workgroup start spread {first.max() - first.min():.0f} ticks, end spread {last.max() - last.min():.0f}; per-workgroup busy {((last - first).mean()):.0f} ticks of span {en - st}')
# first tile vs steady tiles
print('tile index: mean total ticks', [int((d[:, 0, t, 3] - d[:, 0, t, 0])[ok[:, 0, t]].double().mean()) for t in range(int(ntile.max()))])
