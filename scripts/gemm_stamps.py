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
rt = d[:, 0, 31, :].clone()          # per workgroup: s_memrealtime at its first tile's start / last tile's end, s_memtime at the same two points
d[:, :, 31, :] = 0
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
live = [b for b in range(256) if int(ntile[b]) > 0]
first = torch.stack([d[b, 0, 0, 0] for b in live]).double(); last = torch.stack([d[b, 0, int(ntile[b]) - 1, 3] for b in live]).double()
span = float(last.max() - first.min())
q = lambda t, f: float(t.sort().values[int(f * (len(t) - 1))])
print(f'{len(live)} workgroups: start spread {first.max() - first.min():.0f} ticks; finish times after the first start: min {last.min() - first.min():.0f} p50 {q(last, 0.5) - first.min():.0f} '
      f'p90 {q(last, 0.9) - first.min():.0f} max {span:.0f}; per-workgroup busy mean {((last - first).mean()):.0f}; span / event time = {span / ms / 1e6:.3f} GHz if the launch were all span')
# finish time by tile count: the workgroups with one tile more decide the launch
for n in sorted(set(int(x) for x in ntile[live])):
    sel = [i for i, b in enumerate(live) if int(ntile[b]) == n]
    print(f'  {len(sel)} workgroups with {n} tiles: finish p50 {q(last[sel], 0.5) - first.min():.0f} max {last[sel].max() - first.min():.0f}')
# first tile vs steady tiles
print('tile index: mean total ticks', [int((d[:, 0, t, 3] - d[:, 0, t, 0])[ok[:, 0, t]].double().mean()) for t in range(int(ntile.max()))])

lv = rt[:, 1] > 0
rt = rt[lv].double()
clk = ((rt[:, 3] - rt[:, 2]) / (rt[:, 1] - rt[:, 0]) * 100.0)      # MHz: s_memrealtime ticks at 100 MHz
t0r = rt[:, 0].min()
print(f'in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz): median {clk.median():.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})')
print(f'real time: workgroups start within {(rt[:, 0].max() - t0r) / 100:.2f} us, finish {(rt[:, 1].min() - t0r) / 100:.2f} .. {(rt[:, 1].max() - t0r) / 100:.2f} us after the first start '
      f'(median {(rt[:, 1].median() - t0r) / 100:.2f}); event time of the launch {ms * 1e3:.1f} us')
