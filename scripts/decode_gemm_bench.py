#!/usr/bin/env python3
"""The small-batch decode GEMMs of gemm_decode.hip in isolation (development aid): time per launch with the weights rotated over enough
copies to defeat the 256 MB Infinity Cache, knob variants side by side in one process, and a loose check against fp32 torch.
   python scripts/decode_gemm_bench.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E

ROWS = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
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

def rb(t): return t.bfloat16().float()
def rmsnorm_ref(x, gm, eps=1e-5):
    xf = x.float()
    return gm.float() * rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))

W = {'wqkv': copies(QKV, D), 'wo': copies(D, D), 'w13': copies(2 * FF, D), 'w2': copies(D, FF), 'head': copies(V, D, target=1.6e9)}
gm = (1 + 0.1 * torch.randn(D, device=dev, generator=g)).bfloat16()
for M in ROWS:
    x = rnd(M, D)
    ao = rnd(M, D, sc=0.5)
    act = rnd(M, FF, sc=0.5)
    max_tokens = 512
    rope = dict(cos=rnd(max_tokens, 128), sin=rnd(max_tokens, 128), q_out=torch.zeros(M, D, device=dev, dtype=torch.bfloat16),
                kc=torch.zeros(M, 8, max_tokens, 128, device=dev, dtype=torch.bfloat16), vc=torch.zeros(M, 8, max_tokens, 128, device=dev, dtype=torch.bfloat16),
                seqs=torch.arange(M, device=dev, dtype=torch.int32), lens=torch.full((M,), 100, device=dev, dtype=torch.int32), max_tokens=max_tokens)
    act_out = torch.zeros(M, FF, device=dev, dtype=torch.bfloat16)
    logits = torch.zeros(M, V, device=dev, dtype=torch.float32)
    # loose checks (bit equality with the separate kernels is tests/test_gpu_llm.py's job)
    xio = x.clone()
    E.op_decode_gemm(1, W['wo'][0], M, X=ao, xio=xio)
    ref = rb(x.float() + rb(ao.float() @ W['wo'][0].float().t()))
    e1 = float((xio.float() - ref).abs().max())
    E.op_decode_gemm(2, W['w13'][0], M, xres=x, gamma=gm, C_out=act_out)
    h = rb(rmsnorm_ref(x, gm))
    w13 = W['w13'][0].float().reshape(FF // 8, 2, 8, D)
    gt, up = rb(h @ w13[:, 0].reshape(FF, D).t()), rb(h @ w13[:, 1].reshape(FF, D).t())
    ref = rb(rb(torch.nn.functional.silu(gt)) * up)
    e2 = float((act_out.float() - ref).abs().max() / ref.abs().max())
    E.op_decode_gemm(4, W['head'][0], M, xres=x, gamma=gm, C_out=logits)
    ref = rb(h @ W['head'][0].float().t())
    e4 = float((logits - ref).abs().max())
    E.op_decode_gemm(0, W['wqkv'][0], M, xres=x, gamma=gm, rope=rope)
    lin = rb(h @ W['wqkv'][0].float().t()).reshape(M, 8, 6, 128)
    vref = lin[:, :, 5]
    e0 = float((rope['vc'][:, :, 100].float() - vref).abs().max())
    torch.cuda.synchronize()
    print(f'rows {M}: max |d| wo {e1:.3g}, w13 (rel) {e2:.3g}, head {e4:.3g}, wqkv V rows {e0:.3g}')
    res = {}
    SWZ = {nm: [E.op_decode_swizzle(wh, w) for w in W[nm]] for nm, wh in (('wqkv', 0), ('wo', 1), ('w13', 2), ('w2', 3), ('head', 4))}
    # same bits from the decode layout?
    xa, xb = x.clone(), x.clone()
    E.op_decode_gemm(3, W['w2'][0], M, X=act, xio=xa); E.op_decode_gemm(3, W['w2'][0], M, X=act, xio=xb, swizzled=SWZ['w2'][0])
    la, lb = torch.zeros_like(logits), torch.zeros_like(logits)
    E.op_decode_gemm(4, W['head'][0], M, xres=x, gamma=gm, C_out=la); E.op_decode_gemm(4, W['head'][0], M, xres=x, gamma=gm, C_out=lb, swizzled=SWZ['head'][0])
    qa = dict(rope, q_out=torch.zeros_like(rope['q_out']), kc=torch.zeros_like(rope['kc']), vc=torch.zeros_like(rope['vc']))
    qb = dict(rope, q_out=torch.zeros_like(rope['q_out']), kc=torch.zeros_like(rope['kc']), vc=torch.zeros_like(rope['vc']))
    E.op_decode_gemm(0, W['wqkv'][0], M, xres=x, gamma=gm, rope=qa); E.op_decode_gemm(0, W['wqkv'][0], M, xres=x, gamma=gm, rope=qb, swizzled=SWZ['wqkv'][0])
    torch.cuda.synchronize()
    print('   decode layout gives the same bits:', bool(torch.equal(xa, xb) and torch.equal(la, lb) and torch.equal(qa['q_out'], qb['q_out']) and torch.equal(qa['kc'], qb['kc']) and torch.equal(qa['vc'], qb['vc'])))
    for variant in (0, 1):                                   # 0: weights as nn.Linear stores them, 1: the decode layout
        sw = (lambda nm, i: SWZ[nm][i % len(SWZ[nm])]) if variant else (lambda nm, i: None)
        flags = 0
        xs = x.clone()
        res[('wqkv', variant)] = timeit(lambda i: E.op_decode_gemm(0, W['wqkv'][i % len(W['wqkv'])], M, xres=x, gamma=gm, rope=rope, flags=flags, swizzled=sw('wqkv', i)), len(W['wqkv']) * 3)
        res[('wo', variant)] = timeit(lambda i: E.op_decode_gemm(1, W['wo'][i % len(W['wo'])], M, X=ao, xio=xs, flags=flags, swizzled=sw('wo', i)), len(W['wo']) * 3)
        res[('w13', variant)] = timeit(lambda i: E.op_decode_gemm(2, W['w13'][i % len(W['w13'])], M, xres=x, gamma=gm, C_out=act_out, flags=flags, swizzled=sw('w13', i)), len(W['w13']) * 3)
        res[('w2', variant)] = timeit(lambda i: E.op_decode_gemm(3, W['w2'][i % len(W['w2'])], M, X=act, xio=xs, flags=flags, swizzled=sw('w2', i)), len(W['w2']) * 3)
        res[('head', variant)] = timeit(lambda i: E.op_decode_gemm(4, W['head'][i % len(W['head'])], M, xres=x, gamma=gm, C_out=logits, flags=flags, swizzled=sw('head', i)), len(W['head']) * 2)
    # the separate kernels these replace, as far as single ops exist: the K-sliced partial GEMMs / streaming GEMMs alone (no norm, RoPE or add kernel)
    old = {}
    hb = h.bfloat16()
    pq, pd = torch.zeros(8 * M, QKV, device=dev), torch.zeros(8 * M, D, device=dev)
    old['wqkv'] = timeit(lambda i: E.op_gemm(7, hb, W['wqkv'][i % len(W['wqkv'])], kernel=3, out=pq), len(W['wqkv']) * 3)
    old['wo'] = timeit(lambda i: E.op_gemm(7, ao, W['wo'][i % len(W['wo'])], kernel=3, out=pd), len(W['wo']) * 3)
    old['w13'] = timeit(lambda i: E.op_gemm(4, hb, W['w13'][i % len(W['w13'])], kernel=3, out=act_out), len(W['w13']) * 3)
    old['w2'] = timeit(lambda i: E.op_gemm(7, act, W['w2'][i % len(W['w2'])], kernel=3, out=pd), len(W['w2']) * 3)
    old['head'] = timeit(lambda i: E.op_gemm(6, hb, W['head'][i % len(W['head'])], kernel=3, out=logits), len(W['head']) * 2)
    mb = {'wqkv': QKV * D * 2e-6, 'wo': D * D * 2e-6, 'w13': 2 * FF * D * 2e-6, 'w2': D * FF * 2e-6, 'head': V * D * 2e-6}
    for k in ('wqkv', 'wo', 'w13', 'w2', 'head'):
        line = f'  {k:5s} {mb[k]:6.1f} MB: separate GEMM alone {old[k]:7.2f} us ({mb[k] / old[k]:.2f} TB/s) | fused'
        line += f' {res[(k, 0)]:7.2f} ({mb[k] / res[(k, 0)]:.2f} TB/s) | fused, decode layout {res[(k, 1)]:7.2f} ({mb[k] / res[(k, 1)]:.2f} TB/s)'
        print(line, flush=True)
