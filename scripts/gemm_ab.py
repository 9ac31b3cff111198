#!/usr/bin/env python3
"""A/B of gemm256 builds in ONE process (development aid): every library named on the command line is loaded side by side and timed on the same
operands, interleaved rounds, median and min.   python scripts/gemm_ab.py ab/libgbase.so ab/libgkoepi.so ... [-- M,N,K,epi ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import _binding as B

args = sys.argv[1:]
shapes = [(64575, 3072, 1024, 0), (64575, 1024, 1024, 2), (64575, 4096, 1024, 1), (64575, 1024, 4096, 2), (25312, 28672, 4096, 4), (25312, 4096, 14336, 3), (8192, 8192, 8192, 0)]
if '--' in args:
    i = args.index('--')
    shapes = [tuple(int(x) for x in a.split(',')) for a in args[i + 1:]]
    args = args[:i]
libs = []
for path in args:
    lib = C.CDLL(os.path.abspath(path))
    fn = lib.cr_op_gemm
    fn.restype, fn.argtypes = B.SIGNATURES['cr_op_gemm']
    libs.append((os.path.basename(path), fn))
g = torch.Generator(device='cuda').manual_seed(0)
_p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, epi) in shapes:
    nobias = epi >= 100                      # epi + 100: the same epilogue without a bias vector
    epi %= 100
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
    bias = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    scale = (torch.rand(N, device='cuda', generator=g)).bfloat16()
    res = torch.rand(M, N if epi != 4 else N // 2, device='cuda', generator=g).bfloat16() if epi in (2, 3) else None
    Cc = torch.zeros(M, N // 2 if epi == 4 else N, device='cuda', dtype=torch.bfloat16)
    def run(fn):
        rc = fn(epi | (2 << 8), _p(A), K, _p(W), K, _p(Cc), Cc.stride(0), _p(None if nobias else bias), _p(scale) if epi == 2 else _p(None), _p(res), res.stride(0) if res is not None else 0, M, N, K, 0, st())
        assert rc == 0, rc
    n = 10 if 2.0 * M * N * K < 2e12 else 3
    t = {name: [] for name, _ in libs}
    for name, fn in libs:
        run(fn)
    torch.cuda.synchronize()
    for rnd in range(7):
        for name, fn in libs:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(n): run(fn)
            ev[1].record(); torch.cuda.synchronize()
            t[name].append(ev[0].elapsed_time(ev[1]) / n)
    line = f'M={M} N={N} K={K} epi={epi}{" (no bias)" if nobias else ""}:'
    for name, _ in libs:
        v = sorted(t[name])
        line += f'  {name} {v[len(v) // 2]:.4f} ms (min {v[0]:.4f}, {2.0 * M * N * K / v[len(v) // 2] / 1e9:.0f} TF)'
    print(line, flush=True)
