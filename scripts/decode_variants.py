#!/usr/bin/env python3
"""gemm_decode.hip's tuning variants side by side (development aid): the `flags` knob of cr_op_decode_gemm selects a variant per launch (bit 0 forces, bit 1 forbids the RMSNorm
prologue with one wave per row for w1|w3 and the LM head; the default takes it from 3 rows on); every variant must give the bits of flags = 0.   python scripts/decode_variants.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E

ROWS = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5, 8]
FLAGS = [int(x) for x in os.environ.get('FLAGS', '2,1,0').split(',')]
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
D, FF, V, QKV = 4096, 14336, 92553, 6144
def rnd(*s, sc=1.0): return (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
def copies(n, k, target=640e6):
    c = max(2, int(target // (n * k * 2)) + 1)
    return [rnd(n, k, sc=0.02) for _ in range(c)]
def timeit(fn, n_w, reps=6):
    for i in range(n_w): fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for i in range(n_w): fn(i)
        ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / n_w * 1e3)
    return best

W = {'w13': copies(2 * FF, D), 'head': copies(V, D, target=1.6e9), 'wqkv': copies(QKV, D)}
SWZ = {nm: [E.op_decode_swizzle(wh, w) for w in W[nm]] for nm, wh in (('w13', 2), ('head', 4), ('wqkv', 0))}
gm = (1 + 0.1 * torch.randn(D, device=dev, generator=g)).bfloat16()
for M in ROWS:
    x = rnd(M, D)
    max_tokens = 512
    rope_c, rope_s = rnd(max_tokens, 128), rnd(max_tokens, 128)
    def rope():
        return dict(cos=rope_c, sin=rope_s, q_out=torch.zeros(M, D, device=dev, dtype=torch.bfloat16),
                    kc=torch.zeros(M, 8, max_tokens, 128, device=dev, dtype=torch.bfloat16), vc=torch.zeros(M, 8, max_tokens, 128, device=dev, dtype=torch.bfloat16),
                    seqs=torch.arange(M, device=dev, dtype=torch.int32), lens=torch.full((M,), 100, device=dev, dtype=torch.int32), max_tokens=max_tokens)
    ref = None
    for fl in FLAGS:
        a13 = torch.zeros(M, FF, device=dev, dtype=torch.bfloat16)
        E.op_decode_gemm(2, W['w13'][0], M, xres=x, gamma=gm, C_out=a13, flags=fl, swizzled=SWZ['w13'][0])
        lg = torch.zeros(M, V, device=dev, dtype=torch.float32)
        E.op_decode_gemm(4, W['head'][0], M, xres=x, gamma=gm, C_out=lg, flags=fl, swizzled=SWZ['head'][0])
        r = rope()
        E.op_decode_gemm(0, W['wqkv'][0], M, xres=x, gamma=gm, rope=r, flags=fl, swizzled=SWZ['wqkv'][0])
        torch.cuda.synchronize()
        out = (a13.clone(), lg.clone(), r['q_out'], r['kc'], r['vc'])
        if ref is None:
            ref = out
            assert float(a13.float().abs().max()) > 0 and float(lg.abs().max()) > 0
        same = all(bool(torch.equal(a, b)) for a, b in zip(ref, out))
        t2 = timeit(lambda i: E.op_decode_gemm(2, W['w13'][i % len(W['w13'])], M, xres=x, gamma=gm, C_out=a13, flags=fl, swizzled=SWZ['w13'][i % len(W['w13'])]), len(W['w13']) * 3)
        t4 = timeit(lambda i: E.op_decode_gemm(4, W['head'][i % len(W['head'])], M, xres=x, gamma=gm, C_out=lg, flags=fl, swizzled=SWZ['head'][i % len(W['head'])]), len(W['head']) * 2)
        rr = rope()
        t0 = timeit(lambda i: E.op_decode_gemm(0, W['wqkv'][i % len(W['wqkv'])], M, xres=x, gamma=gm, rope=rr, flags=fl, swizzled=SWZ['wqkv'][i % len(W['wqkv'])]), len(W['wqkv']) * 3)
        print(f'rows {M} flags {fl}: same bits as flags {FLAGS[0]}: {same} | wqkv {t0:6.2f} us  w13 {t2:6.2f} us  head {t4:7.2f} us', flush=True)
