"""Per-kernel parity on a real MI355X, through the C ABI (cr_op_*).

References are plain PyTorch fp32 expressions of the same op with the reference
model's bf16 rounding points written out.  Tolerances: outputs are bf16, so one
output ulp is 2^-8 relative; we allow 2 ulp (rtol 1.6e-2) plus an absolute floor
for cancellation.  Integer-valued cases are exact (atol = rtol = 0): they catch
fragment-layout and swizzle mistakes that random data can hide.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1.6e-2


def dev():
    return torch.device('cuda', 0)


def bf(t):
    return t.to(torch.bfloat16)


def rb(t):
    return t.to(torch.bfloat16).float()


@pytest.fixture(scope='module')
def E():
    from callireader_amd import engine
    return engine


@pytest.mark.parametrize('M,N,K', [(300, 256, 64), (128, 128, 128), (1, 128, 64), (257, 384, 192), (1025, 1024, 640)])
def test_gemm_exact_integers(E, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randint(-1, 2, (M, K), generator=g).float()
    W = torch.randint(-1, 2, (N, K), generator=g).float()
    # asymmetric content: make row i of A and row j of W identifiable
    A[:, 0] = (torch.arange(M) % 3 - 1).float()
    W[:, 1] = (torch.arange(N) % 2).float()
    ref = A @ W.t()
    assert ref.abs().max() <= 256
    out = E.op_gemm(0, bf(A).to(dev()), bf(W).to(dev()))
    torch.cuda.synchronize()
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize('M,N,K', [(2049, 520, 128), (2304, 768, 256), (4100, 1024, 640), (2048, 512, 1024), (5000, 300 * 3, 384),
                                   (8300, 2304, 1024), (2100, 520, 4096), (16500, 1280, 256)])
def test_gemm256_exact_integers(E, M, N, K):
    """The 256x256 8-phase kernel, pinned (the dispatcher's cost model would hand several of these shapes to the
    128x128 kernel): exact integer data checks the unit/sub-tile/swizzle maps, the K-tile pairing (K = 128 is a single
    pair), ragged M/N edges, long K loops on data that misses L2, and workgroups that walk several tiles (> 256 tiles:
    the look-ahead across the tile boundary, fragments prefetched over the epilogue)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randint(-1, 2, (M, K), generator=g).float()
    W = torch.randint(-1, 2, (N, K), generator=g).float()
    A[:, 0] = (torch.arange(M) % 3 - 1).float()
    W[:, 1] = (torch.arange(N) % 2).float()
    A[:, K - 1] = ((torch.arange(M) // 7) % 2).float()
    ref = rb(A @ W.t())                                       # integers: the fp32 accumulator is exact, one bf16 rounding
    Ad, Wd = bf(A).to(dev()), bf(W).to(dev())
    outs = [E.op_gemm(0, Ad, Wd, kernel=2) for _ in range(6)]   # repeated launches: a staging race shows as a flaky tile
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o.float().cpu(), ref)


def test_gemm256_random_is_deterministic_and_close(E):
    g = torch.Generator().manual_seed(77)
    M, N, K = 8200, 1024, 4096
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.02)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    ref = rb(A.float() @ W.float().t() + bias.float())
    outs = [E.op_gemm(0, A, W, bias=bias, kernel=2) for _ in range(5)]
    torch.cuda.synchronize()
    torch.testing.assert_close(outs[0].float(), ref, rtol=RTOL, atol=2e-2)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def _rand(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale)


@pytest.mark.parametrize('M,N,K,kern', [(515, 384, 256, 1), (2050, 1024, 1024, 1), (2050, 1024, 1024, 2), (515, 384, 256, 2)])
def test_gemm_epilogues(E, M, N, K, kern):
    g = torch.Generator().manual_seed(1)
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    scale = bf(_rand((N,), g, 0.2) + 0.1).to(dev())
    res = bf(_rand((M, N), g)).to(dev())
    acc = A.float() @ W.float().t()
    lin = rb(acc + bias.float())
    atol = 2e-2

    def close(out, ref):
        torch.cuda.synchronize()
        torch.testing.assert_close(out.float(), ref, rtol=RTOL, atol=atol)

    close(E.op_gemm(0, A, W, bias=bias, kernel=kern), lin)
    close(E.op_gemm(0, A, W, kernel=kern), rb(acc))
    close(E.op_gemm(1, A, W, bias=bias, kernel=kern), rb(torch.nn.functional.gelu(lin)))
    close(E.op_gemm(2, A, W, bias=bias, scale=scale, res=res, kernel=kern), rb(res.float() + rb(lin * scale.float())))
    close(E.op_gemm(3, A, W, bias=bias, res=res, kernel=kern), rb(res.float() + lin))
    out32 = E.op_gemm(6, A, W, out_dtype=torch.float32, kernel=kern)
    close(out32, rb(acc))
    assert out32.dtype == torch.float32
    # in-place residual (C aliases res), as the ViT / LLM residual stream uses it
    x = res.clone()
    from callireader_amd import _binding as B
    from callireader_amd.engine import _p, _stream
    B.check(B.lib.cr_op_gemm(2 | (kern << 8), _p(A), K, _p(W), K, _p(x), N, _p(bias), _p(scale), _p(x), N, M, N, K, 0, _stream()))
    close(x, rb(res.float() + rb(lin * scale.float())))


def test_gemm_epilogue_rounding_is_bit_exact(E):
    """Integer-valued A/W make the fp32 accumulator exact, so every epilogue must reproduce the reference's bf16
    rounding sequence bit for bit (guards against the compiler folding the intermediate roundings away)."""
    g = torch.Generator().manual_seed(12)
    M, N, K = 200, 256, 64
    _bit_exact_epilogues(E, g, M, N, K, 1)


def test_gemm256_epilogue_rounding_is_bit_exact(E):
    _bit_exact_epilogues(E, torch.Generator().manual_seed(13), 2100, 512, 128, 2)


def _bit_exact_epilogues(E, g, M, N, K, kern):
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float()
    bias = bf(_rand((N,), g, 0.37))
    scale = bf(_rand((N,), g, 0.21) + 0.13)
    res = bf(_rand((M, N), g, 1.7))
    acc = A @ W.t()
    lin = rb(acc + bias.float())
    Ad, Wd, bd, sd_, rd = [t.to(dev()) for t in (bf(A), bf(W), bias, scale, res)]

    def same(out, ref):
        torch.cuda.synchronize()
        assert torch.equal(out.float().cpu(), ref)

    same(E.op_gemm(0, Ad, Wd, bias=bd, kernel=kern), lin)
    same(E.op_gemm(2, Ad, Wd, bias=bd, scale=sd_, res=rd, kernel=kern), rb(res.float() + rb(lin * scale.float())))
    same(E.op_gemm(3, Ad, Wd, bias=bd, res=rd, kernel=kern), rb(res.float() + lin))
    same(E.op_gemm(6, Ad, Wd, bias=bd, out_dtype=torch.float32, kernel=kern), lin)
    # GELU/SiLU go through erff/expf whose last bit may differ from the CPU's: allow 1 bf16 ulp there
    out = E.op_gemm(1, Ad, Wd, bias=bd, kernel=kern)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.float().cpu(), rb(torch.nn.functional.gelu(lin)), rtol=2 ** -7, atol=1e-6)


def test_gelu_packed_pairs_equal_the_scalar_form(E):
    """The 256x256 kernel evaluates GELU on pairs with packed fp32 instructions (common.hpp: gelu_erf2), the 128x128 kernel one
    element at a time (gelu_erf): the same bits on every bf16 input, in both positions of a pair and beside any neighbour."""
    bits = torch.arange(0, 65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    x = x[torch.isfinite(x) & (x.abs() < 1e30) & ((x.abs() > 1e-30) | (x == 0))]
    M, N, K = 2048 * ((x.numel() + 2047) // 2048), 512, 128
    A = torch.zeros(M, K)
    A[:x.numel(), 0] = x
    A[:x.numel(), 1] = x.flip(0)                             # the pair partner: another input, of the other sign
    W = torch.zeros(N, K)
    W[0::2, 0] = 1.0                                         # even columns carry x, odd columns its partner
    W[1::2, 1] = 1.0
    a, w = bf(A).to(dev()), bf(W).to(dev())
    o128, o256 = E.op_gemm(1, a, w, kernel=1), E.op_gemm(1, a, w, kernel=2)
    torch.cuda.synchronize()
    assert torch.equal(o128, o256)


@pytest.mark.parametrize('kern', [1, 2])
def test_gelu_epilogue_on_every_bf16_input(E, kern):
    """The GELU epilogue's input is always a bf16 value, so its whole domain is 65 k points: push every normal bf16
    below 1e30 through the epilogue (A = [x, 0, ...], W = e_0: the accumulator IS x) and compare with torch's CPU GELU
    in fp32 rounded to bf16.  The branch-free erf is an fp32-class approximation: like any other fp32 erff it may
    land on the other side of a bf16 rounding boundary for a few inputs of the negative tail, never by more than
    one bf16 step, and nowhere for |x| < 2 (measured: 24 of 65 k inputs, all in [-6, -2])."""
    bits = torch.arange(0, 65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    x = x[torch.isfinite(x) & (x.abs() < 1e30) & ((x.abs() > 1e-30) | (x == 0))]      # no denormals: the matrix core flushes them
    M, N, K = x.numel(), 64, 128
    A = torch.zeros(M, K)
    A[:, 0] = x
    W = torch.zeros(N, K)
    W[:, 0] = 1.0
    out = E.op_gemm(1, bf(A).to(dev()), bf(W).to(dev()), kernel=kern)
    torch.cuda.synchronize()
    got = out[:, 0].cpu()
    assert torch.equal(out[:, 0], out[:, 63])
    ref = torch.nn.functional.gelu(x).to(torch.bfloat16)
    neq = got.float() != ref.float()
    assert int(neq.sum()) <= 64, int(neq.sum())
    assert not bool((neq & (x.abs() < 2)).any())
    step = (got.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    near = neq & (x > -4)
    assert int(step[near].max() if near.any() else 0) <= 1
    # deeper in the tail 1 + erf is a handful of fp32 ulps (the reference's own cancellation): compare absolutely
    far = neq & (x <= -4)
    assert float(((got.float() - ref.float()).abs() / x.abs())[far].max() if far.any() else 0) <= 2.0 ** -22


@pytest.mark.parametrize('M,kern', [(300, 1), (2100, 2)])
def test_gemm_swiglu(E, M, kern):
    g = torch.Generator().manual_seed(2)
    F, K = 512, 256
    A = bf(_rand((M, K), g)).to(dev())
    w1 = bf(_rand((F, K), g, 0.05))
    w3 = bf(_rand((F, K), g, 0.05))
    # interleave [8 rows of w1 | 8 rows of w3] as llm_finalize does
    W = torch.stack([w1.reshape(F // 8, 8, K), w3.reshape(F // 8, 8, K)], dim=1).reshape(2 * F, K).to(dev())
    gte = rb(A.float() @ w1.float().t().to(dev()))
    up = rb(A.float() @ w3.float().t().to(dev()))
    ref = rb(rb(torch.nn.functional.silu(gte)) * up)
    out = E.op_gemm(4, A, W, kernel=kern)
    torch.cuda.synchronize()
    assert out.shape == (M, F)
    torch.testing.assert_close(out.float(), ref, rtol=RTOL, atol=2e-2)


@pytest.mark.parametrize('kern,T,G', [(1, 2, 1024), (2, 2, 1024), (2, 3, 700), (2, 21, 100), (1, 3, 700)])
def test_gemm_patch_rows(E, kern, T, G):
    """G = 1024 is the model's; groups that are no multiple of 8 (700, 100) make an 8-row piece of the 256x256 kernel's four-slices-ahead
    loop straddle two tiles' position rows: those shapes must take the per-row path (round-3 advisor finding)."""
    g = torch.Generator().manual_seed(3)
    N, K = 256, 128
    A = bf(_rand((T * G, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    pos = bf(_rand((G + 1, N), g)).to(dev())
    out = E.op_gemm(5, A, W, bias=bias, res=pos, group=G, kernel=kern)
    torch.cuda.synchronize()
    lin = rb(A.float() @ W.float().t() + bias.float()).reshape(T, G, N)
    ref = rb(lin + pos.float()[1:][None])
    got = out.float().reshape(T, G + 1, N)
    torch.testing.assert_close(got[:, 1:], ref, rtol=RTOL, atol=2e-2)
    assert torch.equal(got[:, 0], torch.zeros(T, N, device=dev()))      # CLS rows are not this kernel's


@pytest.mark.parametrize('epi', [0, 3, 6])
def test_gemm256_ragged_n(E, epi):
    """256x256 kernel on an N that is no multiple of 4 (the vocabulary is 92 553): the last column block takes the
    element-wise epilogue, bias / residual / fp32 output included."""
    g = torch.Generator().manual_seed(40 + epi)
    M, N, K = 2100, 1003, 128
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    res = bf(_rand((M, N), g)).to(dev())
    lin = rb(A.float() @ W.float().t() + bias.float())
    if epi == 6:
        out = E.op_gemm(6, A, W, bias=bias, out_dtype=torch.float32, kernel=2)
        ref = lin
    elif epi == 3:
        out = E.op_gemm(3, A, W, bias=bias, res=res, kernel=2).float()
        ref = rb(res.float() + lin)
    else:
        out = E.op_gemm(0, A, W, bias=bias, kernel=2).float()
        ref = lin
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=RTOL, atol=2e-2)


def test_gemm_ragged_n_and_f32(E):
    g = torch.Generator().manual_seed(4)
    M, N, K = 70, 1003, 128           # N not a multiple of 8 (like vocab 92553)
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    out = E.op_gemm(6, A, W, out_dtype=torch.float32)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, rb(A.float() @ W.float().t()), rtol=RTOL, atol=2e-2)


@pytest.mark.parametrize('M,N,K', [(1, 4096, 4096), (8, 6144, 4096), (17, 4096, 14336), (64, 12288, 1792), (3, 1003, 512), (16, 256, 128)])
def test_gemm_skinny_decode_shapes(E, M, N, K):
    """M <= 64 takes the weight-streaming kernel (4 or 8 K-slices per workgroup)."""
    g = torch.Generator().manual_seed(M + N)
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.03)).to(dev())
    res = bf(_rand((M, N), g)).to(dev())
    acc = A.float() @ W.float().t()
    for out, ref in [(E.op_gemm(0, A, W), rb(acc)), (E.op_gemm(3, A, W, res=res), rb(res.float() + rb(acc))),
                     (E.op_gemm(6, A, W, out_dtype=torch.float32), rb(acc))]:
        torch.cuda.synchronize()
        torch.testing.assert_close(out.float(), ref, rtol=RTOL, atol=2e-2)
    # exact integers: fragment layout / K-slice reduction
    Ai = torch.randint(-1, 2, (M, K), generator=g).float()
    Wi = torch.randint(-1, 2, (N, K), generator=g).float()
    Ai[:, 0] = (torch.arange(M) % 3 - 1).float()
    Wi[:, 1] = (torch.arange(N) % 2).float()
    ref = Ai @ Wi.t()
    if float(ref.abs().max()) <= 256:
        out = E.op_gemm(0, bf(Ai).to(dev()), bf(Wi).to(dev()))
        torch.cuda.synchronize()
        assert torch.equal(out.float().cpu(), ref)


def test_gemm_skinny_swiglu_and_row_independence(E):
    g = torch.Generator().manual_seed(31)
    M, F, K = 9, 1024, 4096
    A = bf(_rand((M, K), g)).to(dev())
    w1 = bf(_rand((F, K), g, 0.03))
    w3 = bf(_rand((F, K), g, 0.03))
    W = torch.stack([w1.reshape(F // 8, 8, K), w3.reshape(F // 8, 8, K)], dim=1).reshape(2 * F, K).to(dev())
    gte = rb(A.float() @ w1.float().t().to(dev()))
    up = rb(A.float() @ w3.float().t().to(dev()))
    out = E.op_gemm(4, A, W)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.float(), rb(rb(torch.nn.functional.silu(gte)) * up), rtol=RTOL, atol=2e-2)
    # a row's result must not depend on the batch it is decoded with (pages decode together)
    one = E.op_gemm(4, A[4:5].contiguous(), W)
    big = E.op_gemm(4, torch.cat([A] * 4).contiguous(), W)
    torch.cuda.synchronize()
    assert torch.equal(one[0], out[4]) and torch.equal(big[9 + 4], out[4])


def test_gemm_skinny_multi_row_tile_path(E):
    """N >= 16384 with M > 8 uses four weight row-tiles per wave; results must equal the one-tile path row for row."""
    g = torch.Generator().manual_seed(41)
    N, K = 16400, 512                      # ragged N (like the 92553-row LM head)
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    A = bf(_rand((40, K), g)).to(dev())
    big = E.op_gemm(6, A, W, out_dtype=torch.float32)                    # M = 40 -> RT = 4
    small = E.op_gemm(6, A[:5].contiguous(), W, out_dtype=torch.float32)  # M = 5  -> RT = 1
    torch.cuda.synchronize()
    assert torch.equal(big[:5], small)
    torch.testing.assert_close(big, rb(A.float() @ W.float().t()), rtol=RTOL, atol=2e-2)
    F = 8192
    w1, w3 = bf(_rand((F, K), g, 0.05)), bf(_rand((F, K), g, 0.05))
    Wi = torch.stack([w1.reshape(F // 8, 8, K), w3.reshape(F // 8, 8, K)], dim=1).reshape(2 * F, K).to(dev())
    o_big = E.op_gemm(4, A, Wi)
    o_small = E.op_gemm(4, A[:7].contiguous(), Wi)
    torch.cuda.synchronize()
    assert torch.equal(o_big[:7], o_small)
    gte, up = rb(A.float() @ w1.float().t().to(dev())), rb(A.float() @ w3.float().t().to(dev()))
    torch.testing.assert_close(o_big.float(), rb(rb(torch.nn.functional.silu(gte)) * up), rtol=RTOL, atol=2e-2)


def test_gemm_stream_kernel_gives_the_k_split_kernels_bits(E, monkeypatch):
    """More than 8 rows at N > 8192, K % 2048 == 0 (w1|w3, the LM head of a decode batch) take gemm_stream.hip: X shared through LDS,
    every wave walking the whole K range as the four quarters the K-split kernel gives its four waves, summed in the same order.
    Its results must equal the K-split kernel's bit for bit: the reference side sends the same rows in chunks of <= 8, which
    gemm_stream_supported() refuses, so they really run gemm_skinny.hip's K-split kernel (round-4 advice: chunks of 16 had started to
    take the stream kernel too) -- and match the reference arithmetic; ragged N (92 553-row vocabulary), every row count class incl. 9 and 16."""
    g = torch.Generator().manual_seed(43)
    N, K = 16400 + 9, 2048
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    CH = 8                                                                     # rows per reference launch: <= 8 rows never take the stream kernel
    for M in (9, 16, 17, 33, 64):
        A = bf(_rand((M, K), g)).to(dev())
        big = E.op_gemm(6, A, W, bias=bias, out_dtype=torch.float32)           # stream kernel
        small = torch.cat([E.op_gemm(6, A[i:i + CH].contiguous(), W, bias=bias, out_dtype=torch.float32) for i in range(0, M, CH)])   # K-split kernel
        torch.cuda.synchronize()
        assert torch.equal(big, small)
        torch.testing.assert_close(big, rb(A.float() @ W.float().t() + bias.float()), rtol=RTOL, atol=2e-2)
        res = bf(_rand((M, N), g)).to(dev())
        o3 = E.op_gemm(3, A, W, bias=bias, res=res)
        o3s = torch.cat([E.op_gemm(3, A[i:i + CH].contiguous(), W, bias=bias, res=res[i:i + CH].contiguous()) for i in range(0, M, CH)])
        o0 = E.op_gemm(0, A, W)
        o0s = torch.cat([E.op_gemm(0, A[i:i + CH].contiguous(), W) for i in range(0, M, CH)])
        torch.cuda.synchronize()
        assert torch.equal(o3, o3s) and torch.equal(o0, o0s)
    F, K = 14336, 4096                                                         # w1|w3 itself
    w1, w3 = bf(_rand((F, K), g, 0.03)), bf(_rand((F, K), g, 0.03))
    Wi = torch.stack([w1.reshape(F // 8, 8, K), w3.reshape(F // 8, 8, K)], dim=1).reshape(2 * F, K).to(dev())
    A = bf(_rand((64, K), g)).to(dev())
    big = E.op_gemm(4, A, Wi)
    small = torch.cat([E.op_gemm(4, A[i:i + 8].contiguous(), Wi) for i in range(0, 64, 8)])
    torch.cuda.synchronize()
    assert torch.equal(big, small)
    gte, up = rb(A.float() @ w1.float().t().to(dev())), rb(A.float() @ w3.float().t().to(dev()))
    torch.testing.assert_close(big.float(), rb(rb(torch.nn.functional.silu(gte)) * up), rtol=RTOL, atol=2e-2)


@pytest.mark.parametrize('M,N,K', [(32, 6144, 4096), (40, 4096, 14336), (7, 4096, 4096), (64, 256, 512), (64, 6144, 4096)])
def test_gemm_skinny_k_sliced_partials(E, M, N, K):
    """Decode's wqkv / wo / w2: fp32 partial sums of K-slices (tall workgroups), summed by the consumer.  The slices
    add up to the product, and a row's slices do not depend on the rows it is batched with."""
    g = torch.Generator().manual_seed(M + N)
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.03)).to(dev())
    part = E.op_gemm(7, A, W).reshape(8, M, N)
    few = E.op_gemm(7, A[:5].contiguous(), W).reshape(8, 5, N)
    torch.cuda.synchronize()
    assert torch.equal(part[:, :5], few)
    used = int((part.abs().amax(dim=(1, 2)) > 0).sum())
    assert 1 <= used <= 8 and float(part[used:].abs().max() if used < 8 else 0) == 0.0
    total = torch.zeros(M, N, device=dev())
    for s_ in range(used):
        total = total + part[s_]
    torch.testing.assert_close(total, A.float() @ W.float().t(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize('M,N,K,kern', [(300, 1000, 128, 1), (2304, 1000, 128, 2), (2100, 4100, 256, 2), (70, 130, 64, 1)])
def test_gemm_row_argmax_partials(E, M, N, K, kern):
    """EPI_ARGMAX (cosine VQ, models/similarity.py:19-21): per row and 64-column block the first maximum of bf16(A.W^T).
    Small-integer data make the products exact and full of ties, so the first-index rule is what is being tested;
    ragged N, ragged M."""
    g = torch.Generator().manual_seed(40 + M)
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float()
    W[7] = W[3]                                                    # duplicated table rows: equal similarity, index 3 must win over 7
    W[N - 1] = W[N - 2]
    out = E.op_gemm(8, bf(A).to(dev()), bf(W).to(dev()), kernel=kern)
    torch.cuda.synchronize()
    sim = rb(A @ W.t())
    nblk = (N + 63) // 64
    part = out[:, :nblk].cpu()
    val = ((part >> 32) & 0xFFFFFFFF).to(torch.int32).view(torch.float32)
    col = (part & 0xFFFFFFFF).to(torch.int64)
    for b in range(nblk):
        blk = sim[:, b * 64:(b + 1) * 64]
        mx, am = blk.max(dim=1)                                    # torch: first maximal index
        assert torch.equal(val[:, b], mx), b
        assert torch.equal(col[:, b], am + b * 64), b


@pytest.mark.parametrize('M,N,K', [(32800, 1024, 1024), (32800, 1024, 4096), (32800, 3072, 1024), (2100, 512, 256)])
def test_gemm_tail_rows_take_the_small_kernel(E, M, N, K):
    """BASELINE config 2 (32 tiles) is 128 x 256 + 32 rows: the dispatcher cuts the 32 rows off the 256x256 launch (they would cost a
    whole round of tiles) and sends them through the 128x128 kernel.  A row's result must not depend on which kernel made it:
    random data, every epilogue of the ViT layer, dispatcher's choice against the pinned un-split 256x256 launch, bit for bit."""
    g = torch.Generator().manual_seed(M + K)
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.03)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    scale = bf(_rand((N,), g, 0.2) + 0.1).to(dev())
    res = bf(_rand((M, N), g)).to(dev())
    for epi, kw in ((0, dict(bias=bias)), (1, dict(bias=bias)), (2, dict(bias=bias, scale=scale, res=res)), (3, dict(res=res))):
        auto = E.op_gemm(epi, A, W, **kw)
        pinned = E.op_gemm(epi, A, W, kernel=2, **kw)
        small = E.op_gemm(epi, A[-300:], W, kernel=1, **{k: (v[-300:] if k == 'res' else v) for k, v in kw.items()})
        torch.cuda.synchronize()
        assert torch.equal(auto, pinned), epi
        assert torch.equal(small, pinned[-300:]), epi          # and the 128x128 kernel agrees on rows well inside a 256-row tile too


def test_gemm_rejects_bad_k(E):
    A = torch.zeros(16, 72, device=dev(), dtype=torch.bfloat16)
    W = torch.zeros(16, 72, device=dev(), dtype=torch.bfloat16)
    from callireader_amd._binding import CalliReaderError
    with pytest.raises(CalliReaderError):
        E.op_gemm(0, A, W)


@pytest.mark.parametrize('rows,n', [(1025 * 2, 1024), (5, 1024), (777, 4096)])
def test_layernorm(E, rows, n):
    g = torch.Generator().manual_seed(5)
    x = bf(_rand((rows, n), g) * 2 + 0.3).to(dev())
    gamma = bf(_rand((n,), g, 0.1) + 1).to(dev())
    beta = bf(_rand((n,), g, 0.1)).to(dev())
    out = E.op_layernorm(x, gamma, beta, 1e-6)
    torch.cuda.synchronize()
    ref = rb(torch.nn.functional.layer_norm(x.float(), (n,), gamma.float(), beta.float(), 1e-6))
    torch.testing.assert_close(out.float(), ref, rtol=RTOL, atol=1e-2)


def test_layernorm_pixel_shuffle(E):
    from oracle import vision
    g = torch.Generator().manual_seed(6)
    T = 3
    v = bf(_rand((T, 1025, 1024), g))
    gamma = bf(_rand((4096,), g, 0.1) + 1)
    beta = bf(_rand((4096,), g, 0.1))
    out = E.op_layernorm(v.to(dev()), gamma.to(dev()), beta.to(dev()), 1e-5, pixel_shuffle=True)
    torch.cuda.synchronize()
    x = v[:, 1:, :].reshape(T, 32, 32, 1024)
    x = vision.pixel_shuffle(x, 0.5).reshape(T * 256, 4096)
    ref = rb(torch.nn.functional.layer_norm(x.float(), (4096,), gamma.float(), beta.float(), 1e-5))
    torch.testing.assert_close(out.float().cpu(), ref, rtol=RTOL, atol=1e-2)


def test_rmsnorm(E):
    g = torch.Generator().manual_seed(7)
    x = bf(_rand((333, 4096), g) * 3).to(dev())
    w = bf(_rand((4096,), g, 0.1) + 1).to(dev())
    out = E.op_rmsnorm(x, w, 1e-5)
    torch.cuda.synchronize()
    xf = x.float()
    h = rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5))
    torch.testing.assert_close(out.float(), rb(w.float() * h), rtol=RTOL, atol=1e-2)


def _attn_ref(q, k, v, causal, q_pos0, prescale, s_div):
    """q (B,H,Sq,D), k/v (B,H,Sk,D) float tensors holding bf16 values; reference rounding points."""
    if prescale != 1.0:
        q = rb(q * prescale)
    s = rb(q @ k.transpose(-1, -2))
    if s_div != 1.0:
        s = rb(s / s_div)
    if causal:
        Sq, Sk = s.shape[-2:]
        qpos = q_pos0 + torch.arange(Sq, device=s.device)[:, None]
        s = s.masked_fill(torch.arange(Sk, device=s.device)[None, :] > qpos, float('-inf'))
    p = rb(torch.softmax(s, dim=-1))
    return rb(p @ v)


def test_attention_vit_shape(E):
    g = torch.Generator().manual_seed(8)
    Bn, S, H, D = 2, 1025, 16, 64
    qkv = bf(_rand((Bn, S, 3 * H * D), g)).to(dev())
    o = torch.zeros(Bn, S, H * D, device=dev(), dtype=torch.bfloat16)
    C3, C1 = 3 * H * D, H * D
    E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o,
                   [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
    torch.cuda.synchronize()
    t = qkv.float().reshape(Bn, S, 3, H, D).permute(2, 0, 3, 1, 4)
    ref = _attn_ref(t[0], t[1], t[2], False, 0, 0.125, 1.0).transpose(1, 2).reshape(Bn, S, C1)
    torch.testing.assert_close(o.float(), ref, rtol=RTOL, atol=1.5e-2)


def test_attention_vit_shape_far_above_the_reference_point(E):
    """attention_vit.hip keeps the reference point of its exponentials where the CLS key put it and never rescales inside the sweep; a
    query whose scores lie far above that point (here +256: exp2 overflows, the row sum is not finite) must send its whole block
    through the exact rescaling sweep.  Head 0 forces it for a few queries of several blocks, head 1 is ordinary data: both must match
    the reference arithmetic, CLS query row included."""
    g = torch.Generator().manual_seed(12)
    Bn, S, H, D = 1, 1025, 2, 64
    qkv = bf(_rand((Bn, S, 3 * H * D), g) * 0.5)
    C3, C1 = 3 * H * D, H * D
    spike = torch.full((D,), 4.0)
    for qrow in (0, 5, 130, 131, 700, 1024):                 # the CLS query and patch queries of blocks 0, 1, 5, 7 (head 0)
        qkv[0, qrow, 0:D] = spike                            # q . k = 0.125 * 4 * 4 * 64 = 128 against the spiked keys
    qkv[0, 0, C1:C1 + D] = -spike                            # CLS key: score -128 for those queries = their reference point
    for krow in (3, 400, 1000):
        qkv[0, krow, C1:C1 + D] = spike                      # patch keys 256 above it
    qkv = qkv.to(dev())
    o = torch.zeros(Bn, S, C1, device=dev(), dtype=torch.bfloat16)
    E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o,
                   [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D], Bn, H, S, S, D, q_prescale=0.125)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    t = qkv.float().reshape(Bn, S, 3, H, D).permute(2, 0, 3, 1, 4)
    ref = _attn_ref(t[0], t[1], t[2], False, 0, 0.125, 1.0).transpose(1, 2).reshape(Bn, S, C1)
    torch.testing.assert_close(o.float(), ref, rtol=RTOL, atol=1.5e-2)


def test_attention_vit_launches_back_to_back_are_identical(E, monkeypatch):
    """Every launch of attention_vit.hip on the same input gives the same bits, and they stay within one bf16 step of the generic
    kernel's.  Launches go out back to back (no synchronisation in between): with four workgroups per CU a wave used to leave the
    tile's closing barrier with its last V fragment reads still queued in the LDS, behind a sibling's fill of that very buffer -- one
    wave's d >= 32 columns of one tile's share wrong, once in a few launches (8 launches x 3 rounds of 96 tiles caught it every time)."""
    g = torch.Generator(device='cuda').manual_seed(3)
    Bn, S, H, D = 96, 1025, 16, 64
    C3, C1 = 3 * H * D, H * D
    qkv = torch.randn(Bn, S, C3, device=dev(), generator=g).bfloat16()
    strides = [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D]

    def launch(o):
        E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, strides, Bn, H, S, S, D, q_prescale=0.125)
    monkeypatch.setenv('CR_VIT_ATTN', '0')
    ref = torch.zeros(Bn, S, C1, device=dev(), dtype=torch.bfloat16)
    launch(ref)
    torch.cuda.synchronize()
    monkeypatch.setenv('CR_VIT_ATTN', '1')
    for _ in range(3):
        outs = [torch.full((Bn, S, C1), 7.0, device=dev(), dtype=torch.bfloat16) for _ in range(8)]
        for o in outs:
            launch(o)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, outs[0])
        assert float((outs[0].float() - ref.float()).abs().max()) <= 2 ** -7


def test_attention_vit_two_wave_form_gives_the_four_wave_forms_bits(E, monkeypatch):
    """attention_vit.hip, round 5: workgroups of TWO waves with 64 queries each (CR_VIT_ATTN_NW=2: every K / V fragment read from LDS feeds two MFMAs)
    beside the four-wave form.  Per query the arithmetic is the same instruction sequence, so the outputs must be the same bits -- ordinary data (CLS row
    and its partial / combine path included), the spiked data that sends blocks through the exact re-sweep, and back-to-back launches as a race screen."""
    g = torch.Generator(device='cuda').manual_seed(5)
    Bn, S, H, D = 40, 1025, 16, 64
    C3, C1 = 3 * H * D, H * D
    qkv = torch.randn(Bn, S, C3, device=dev(), generator=g).bfloat16()
    qkv[1, 5, 0:D] = 4.0; qkv[1, 0, C1:C1 + D] = -4.0; qkv[1, 400, C1:C1 + D] = 4.0       # tile 1, head 0: a flagged block (exact re-sweep)
    strides = [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D]

    def launch(o):
        E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, strides, Bn, H, S, S, D, q_prescale=0.125)
    monkeypatch.setenv('CR_VIT_ATTN_NW', '4')
    ref = torch.zeros(Bn, S, C1, device=dev(), dtype=torch.bfloat16)
    launch(ref)
    torch.cuda.synchronize()
    monkeypatch.setenv('CR_VIT_ATTN_NW', '2')
    for _ in range(3):
        outs = [torch.full((Bn, S, C1), 7.0, device=dev(), dtype=torch.bfloat16) for _ in range(6)]
        for o in outs:
            launch(o)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, ref)
    assert torch.isfinite(ref.float()).all()


def test_decode_attention_kernels_against_the_reference_arithmetic(E):
    """The batched decode attention alone (cr_op_decode_attention): one new token per row over a cache of ragged lengths -- a single key, lengths on both sides of the 64-key
    wave, the 256-key split and the 4-key lane groups, a long context (5 000 keys: 20 splits), rows in arbitrary cache slots.  Both kernels -- attention_decode.hip's streaming
    vector-pipe form (the product) and attention.hip's matrix-core split form -- against the reference's rounding points (modeling_internlm2.py:393-410: scores bf16, / sqrt(128)
    -> bf16, fp32 softmax, bf16 probabilities); the two kernels within one bf16 step of each other; a row's result independent of the batch it is launched in (same bits)."""
    g = torch.Generator().manual_seed(21)
    lens = [0, 2, 3, 62, 63, 64, 254, 255, 256, 257, 999, 4999]            # keys = lens + 1
    slots = [5, 0, 7, 3, 11, 1, 9, 2, 10, 4, 8, 6]
    n_slots, max_tokens, H, G, D = 12, 5008, 8, 4, 128
    kc = bf(_rand((n_slots, H, max_tokens, D), g)).to(dev())
    vc = bf(_rand((n_slots, H, max_tokens, D), g)).to(dev())
    q = bf(_rand((len(lens), H * G * D), g) * 0.7).to(dev())
    seqs = torch.tensor(slots, dtype=torch.int32, device=dev())
    lens_d = torch.zeros(n_slots, dtype=torch.int32, device=dev())
    lens_d[seqs.long()] = torch.tensor(lens, dtype=torch.int32, device=dev())
    out = E.op_decode_attention(q, kc, vc, seqs, lens_d)
    out_mc = E.op_decode_attention(q, kc, vc, seqs, lens_d, which=1)
    torch.cuda.synchronize()
    sdiv = math.sqrt(128.0)
    for b, (L, slot) in enumerate(zip(lens, slots)):
        qq = q[b].float().reshape(H, G, D)
        k = kc[slot, :, :L + 1].float()
        v = vc[slot, :, :L + 1].float()
        ref = _attn_ref(qq, k, v, False, 0, 1.0, sdiv).reshape(-1)
        torch.testing.assert_close(out[b].float(), ref, rtol=RTOL, atol=1.5e-2)
        torch.testing.assert_close(out_mc[b].float(), ref, rtol=RTOL, atol=1.5e-2)
    assert float((out.float() - out_mc.float()).abs().max()) <= 2 ** -6
    # batch independence: the last row alone, and the rows in another order, give the same bits
    alone = E.op_decode_attention(q[-1:].contiguous(), kc, vc, seqs[-1:].contiguous(), lens_d)
    perm = torch.tensor([3, 11, 0, 7, 5, 1, 9, 2, 10, 4, 8, 6], device=dev())
    shuffled = E.op_decode_attention(q[perm].contiguous(), kc, vc, seqs[perm].contiguous(), lens_d)
    torch.cuda.synchronize()
    assert torch.equal(alone[0], out[-1]) and torch.equal(shuffled, out[perm])


def test_attention_exact_identity_layout(E):
    """V = one-hot rows, uniform scores: output row = mean of V rows -> exact in bf16; catches V^T/tr-read mistakes."""
    Bn, S, H, D = 1, 64, 1, 64
    q = torch.zeros(Bn, S, H * D, device=dev(), dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    v = torch.zeros_like(q)
    for key in range(S):
        v[0, key, key % D] = float(key % 7 + 1) * 64.0     # asymmetric in (key, d)
    o = torch.zeros_like(q)
    E.op_attention(q, k, v, o, [S * D, D, D] * 4, Bn, H, S, S, D)
    torch.cuda.synchronize()
    ref = v.float().mean(dim=1, keepdim=True).expand(-1, S, -1)
    torch.testing.assert_close(o.float(), rb(ref), rtol=0, atol=0)


@pytest.mark.parametrize('Sq,Sk,q_pos0', [(300, 300, 0), (1, 77, 76), (130, 200, 70)])
def test_attention_llm_causal_gqa(E, Sq, Sk, q_pos0):
    g = torch.Generator().manual_seed(9 + Sq)
    H, KV, D = 8, 2, 128
    q = bf(_rand((Sq, H * D), g)).to(dev())
    k = bf(_rand((KV, Sk, D), g)).to(dev())
    v = bf(_rand((KV, Sk, D), g)).to(dev())
    o = torch.zeros(Sq, H * D, device=dev(), dtype=torch.bfloat16)
    E.op_attention(q, k, v, o, [0, H * D, D, 0, D, Sk * D, 0, D, Sk * D, 0, H * D, D], 1, H, Sq, Sk, D,
                   kv_group=H // KV, causal=True, q_pos0=q_pos0, s_div=math.sqrt(D))
    torch.cuda.synchronize()
    qf = q.float().reshape(Sq, H, D).permute(1, 0, 2)[None]
    kf = k.float().repeat_interleave(H // KV, dim=0)[None]
    vf = v.float().repeat_interleave(H // KV, dim=0)[None]
    ref = _attn_ref(qf, kf, vf, True, q_pos0, 1.0, math.sqrt(D))[0].permute(1, 0, 2).reshape(Sq, H * D)
    torch.testing.assert_close(o.float(), ref, rtol=RTOL, atol=1.5e-2)


def test_attention_softmax_spike(E):
    """One key dominates late in the sequence: forces the online-softmax rescale branch."""
    g = torch.Generator().manual_seed(11)
    Bn, S, H, D = 1, 512, 1, 64
    q = bf(_rand((Bn, S, D), g)).to(dev())
    k = bf(_rand((Bn, S, D), g)).to(dev())
    v = bf(_rand((Bn, S, D), g)).to(dev())
    k[0, 400] = q[0, 5] * 4.0
    o = torch.zeros_like(q)
    E.op_attention(q, k, v, o, [S * D, D, D] * 4, Bn, H, S, S, D)
    torch.cuda.synchronize()
    ref = _attn_ref(q.float()[:, None], k.float()[:, None], v.float()[:, None], False, 0, 1.0, 1.0)[:, 0]
    torch.testing.assert_close(o.float(), ref, rtol=RTOL, atol=1.5e-2)


@pytest.mark.parametrize('M,N,K', [(2304, 768, 256), (4100, 1024, 640 + 128), (5000, 2304, 1024), (2049, 512, 128)])
def test_gemm256_both_schedules_give_the_same_bits(E, M, N, K):
    """The 256x256 kernel has two schedules of the same sums (16- and 32-MFMA matrix slots, round 4); the launcher picks one per shape, so a
    shape's bits may not depend on the pick: every epilogue that has both instances, random data, exact equality, plus the integer-exact
    reference for the plain store."""
    g = torch.Generator().manual_seed(900 + M)
    A = bf(_rand((M, K), g)).to(dev())
    W = bf(_rand((N, K), g, 0.05)).to(dev())
    bias = bf(_rand((N,), g, 0.1)).to(dev())
    scale = bf(_rand((N,), g)).to(dev())
    res = bf(_rand((M, N), g)).to(dev())
    for epi, kw in ((0, dict(bias=bias)), (1, dict(bias=bias)), (2, dict(bias=bias, scale=scale, res=res)), (3, dict(res=res)), (6, dict(bias=bias, out_dtype=torch.float32))):
        a = E.op_gemm(epi, A, W, kernel=5, **kw)
        b = E.op_gemm(epi, A, W, kernel=6, **kw)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (epi, float((a.float() - b.float()).abs().max()))
    if N % 16 == 0:
        a = E.op_gemm(4, A, W, kernel=5)
        b = E.op_gemm(4, A, W, kernel=6)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    Ai = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16).to(dev())
    Wi = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16).to(dev())
    ref = (Ai.float() @ Wi.float().t()).to(torch.bfloat16)
    for kern in (5, 6):
        assert torch.equal(E.op_gemm(0, Ai, Wi, kernel=kern), ref), kern


@pytest.mark.parametrize('epi', [1, 2])
def test_gemm256_launches_back_to_back_are_identical(E, epi):
    """The persistent 256x256 kernel keeps LDS-DMA in flight across raw barriers and (GELU) reads its table beside them: the same
    launch issued back to back must give the same bits every time."""
    g = torch.Generator(device='cuda').manual_seed(5)
    M, N, K = 64575, 1024, 1024
    A = (torch.rand(M, K, device=dev(), generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device=dev(), generator=g) * 2 - 1) * 0.05).bfloat16()
    bias = (torch.rand(N, device=dev(), generator=g) * 0.1).bfloat16()
    scale = (torch.rand(N, device=dev(), generator=g) * 0.1).bfloat16()
    res = torch.rand(M, N, device=dev(), generator=g).bfloat16()
    kw = dict(bias=bias, kernel=2) if epi == 1 else dict(bias=bias, scale=scale, res=res, kernel=2)
    for _ in range(3):
        outs = [E.op_gemm(epi, A, W, **kw) for _ in range(8)]
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, outs[0])


@pytest.mark.parametrize('M', [9, 16, 33, 64])
def test_weight_streaming_gemms_take_the_decode_layout_with_the_same_bits(E, M):
    """The decode GEMMs of 9..64 rows (K-sliced partial sums for wqkv / wo / w2, SwiGLU for w1|w3, fp32 logits with a ragged vocabulary) read the
    decode-layout copy of their weight (cr_op_decode_swizzle: a contiguous KiB per load instruction; wqkv in its RoPE tile order, the sums still
    landing in their nn.Linear columns): every output is the same bits as from the nn.Linear layout."""
    g = torch.Generator().manual_seed(M)
    D, FF, V, QKV = 4096, 14336, 8201, 6144
    def w(n, k): return bf(_rand((n, k), g, 0.02)).to(dev())
    x, act = bf(_rand((M, D), g)).to(dev()), bf(_rand((M, FF), g, 0.5)).to(dev())
    for which, epi, A, W, kind in ((0, 7, x, w(QKV, D), 2), (1, 7, x, w(D, D), 1), (2, 4, x, w(2 * FF, D), 1), (3, 7, act, w(D, FF), 1), (4, 6, x, w(V, D), 1)):
        S = E.op_decode_swizzle(which, W)
        kw = dict(out_dtype=torch.float32) if epi == 6 else {}
        a = E.op_gemm(epi, A, W, kernel=3, **kw)
        b = E.op_gemm(epi, A, W, kernel=3, decode_layout=(kind, S), **kw)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (which, M)
        assert float(a.float().abs().max()) > 0
