"""fp8 (e4m3) weight path of the batched decode -- BASELINE.json configs[4]'s option; the reference has no fp8, so the gate
is agreement with this engine's own bf16 path, which is what the rest of the suite pins to the reference.

  * the row quantiser equals torch's float8_e4m3fn rounding of w / scale, scale = max|w| / 448;
  * the weight-streaming GEMM on e4m3 weights is EXACT on data that e4m3 represents (fragment pairing of the 64-deep steps,
    K-slices, SwiGLU row interleave, row scales);
  * a 2-layer InternLM2 at full width: teacher-forced decode logits of the fp8 path against the bf16 path -- tolerance
    stated below -- and the greedy picks' agreement.
Tolerance: e4m3 keeps 3 mantissa bits, so every weight moves by up to 2^-4 of itself (rms 2^-4 / sqrt 3 = 3.6 %), whatever the
row scale.  With random-init weights the terms of a dot product are independent, the sum and its error both grow as sqrt(K),
and EVERY linear layer's output carries ~3.6 % of relative noise; this test's model stacks nine linears (2 layers x 4 + the
LM head) in front of the logits: measured rel-L2 1.1e-1.  Bound: 1.5e-1.  (Random weights are the unfavourable case: nothing
averages out.  For scale: the reference's own bf16-vs-fp32 logit difference at full depth is 8.3e-2, tests/golden/full_depth.npz.)"""
import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def E():
    from callireader_amd import engine
    return engine


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def test_row_quantiser_equals_torch_e4m3(E):
    g = torch.Generator().manual_seed(0)
    W = (torch.randn(300, 1024, generator=g) * 0.02).bfloat16()
    W[5] = 0                                                   # an all-zero row keeps scale 1
    W[6, 3] = 1.5                                              # an outlier sets its row's scale
    q, sc = E.op_quantize_fp8(W.cuda())
    torch.cuda.synchronize()
    mx = W.float().abs().amax(dim=1)
    exp_sc = torch.where(mx > 0, mx / 448.0, torch.ones_like(mx))
    assert torch.equal(sc.cpu(), exp_sc)
    exp_q = (W.float() * (1.0 / exp_sc)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), exp_q)
    assert int(q[6, 3]) == 0x7E                                # +448, the largest finite e4m3


@pytest.mark.parametrize('M,N,K,epi', [(1, 4096, 4096, 0), (33, 6144, 4096, 7), (64, 4096, 14336, 7), (17, 1024, 512, 3), (40, 2048, 1024, 4), (5, 9000, 4096, 6),
                                       (33, 16409, 2048, 6), (64, 28672, 4096, 4), (9, 9008, 4096, 4)])      # the last three: gemm_stream.hip's e4m3 form (9..64 rows, N > 8192)
def test_fp8_weight_gemm_is_exact_on_representable_data(E, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N)
    W = torch.randint(-7, 8, (N, K), generator=g).float()
    W[:, 0] = 7.0                                              # every row's maximum is 7 -> scale 1/64, bytes = 64 w exactly
    X = torch.randint(-2, 3, (M, K), generator=g).float()
    q, sc = E.op_quantize_fp8(W.bfloat16().cuda())
    assert torch.equal(sc.cpu(), torch.full((N,), 7.0 / 448.0))
    res = torch.randint(-3, 4, (M, N), generator=g).float().bfloat16() if epi == 3 else None
    out = E.op_gemm_fp8(epi, X.bfloat16().cuda(), q, sc, res=res.cuda() if res is not None else None,
                        out_dtype=torch.float32 if epi == 6 else torch.bfloat16)
    # the e4m3 copy's decode layout (round 5: cr_enable_fp8_decode keeps it; one contiguous KiB per load instruction): the same values in the same
    # registers, so the same bits -- ragged N included (5, 9000)
    out_dl = E.op_gemm_fp8(epi, X.bfloat16().cuda(), q, sc, res=res.cuda() if res is not None else None,
                           out_dtype=torch.float32 if epi == 6 else torch.bfloat16, decode_layout=E.op_decode_swizzle8(q))
    torch.cuda.synchronize()
    assert torch.equal(out_dl, out)
    if N > 8192 and M > 8 and epi in (4, 6):
        # ... and the stream kernel's e4m3 form gives the K-split kernel's bits (chunks of <= 8 rows never take the stream kernel)
        chunks = torch.cat([E.op_gemm_fp8(epi, X[i:i + 8].bfloat16().cuda().contiguous(), q, sc, out_dtype=torch.float32 if epi == 6 else torch.bfloat16) for i in range(0, M, 8)])
        torch.cuda.synchronize()
        assert torch.equal(chunks, out)
    acc = X @ W.t()
    if epi == 7:                                               # fp32 K-slices: their sum is the product
        got = out.reshape(8, M, N).sum(dim=0).cpu()
        assert torch.equal(got, acc)
    elif epi == 4:                                             # rows [8 gate | 8 up] per 16 -> silu(gate) * up
        a3 = acc.reshape(M, N // 16, 16)
        gate, up = a3[:, :, :8].reshape(M, -1).bfloat16().float(), a3[:, :, 8:].reshape(M, -1).bfloat16().float()
        ref = (torch.nn.functional.silu(gate).bfloat16().float() * up).bfloat16().float()
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2 ** -7, atol=1e-6)
    elif epi == 3:
        assert torch.equal(out.float().cpu(), (res.float() + acc.bfloat16().float()).bfloat16().float())
    else:
        assert torch.equal(out.float().cpu(), acc.bfloat16().float())


def test_fp8_decode_against_the_bf16_path():
    from callireader_amd.engine import Engine
    dims = ModelDims.reduced(vit_layers=1, llm_layers=2, rs_depth=1, vocab=8201)
    sd = synthetic.make_state_dict(dims, parts=('llm',), seed=0)
    eng = Engine(dims, max_pos=1024)
    eng.load_state_dict(sd)
    eng.load_rope()
    eng.finalize()
    g = torch.Generator().manual_seed(5)
    prompts = [(torch.randn(1, S, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda() for S in (200, 77, 130)]
    steps = 12

    def run(fp8):
        eng.enable_fp8_decode(fp8)
        kv = eng.kv_alloc(3, 512)
        eng.prefill_batch(kv, [0, 1, 2], prompts)              # prefill stays bf16 either way
        logits = []
        for _ in range(steps):
            logits.append(eng.decode(kv, [0, 1, 2], want_logits=True).float().cpu())
        ids = [kv.generated(i) for i in range(3)]
        kv.free()
        return logits, ids
    ref_logits, ref_ids = run(False)
    # teacher-forced comparison: the fp8 run is fed the bf16 run's ids so that every step compares like with like
    eng.enable_fp8_decode(True)
    kv = eng.kv_alloc(3, 512)
    eng.prefill_batch(kv, [0, 1, 2], prompts)
    worst, agree, total = 0.0, 0, 0
    for t in range(steps):
        force = torch.tensor([ref_ids[i][t] for i in range(3)])
        lg = eng.decode(kv, [0, 1, 2], force_tokens=force, want_logits=True).float().cpu()
        worst = max(worst, rel_l2(lg, ref_logits[t]))
        agree += int((lg.argmax(dim=1) == ref_logits[t].argmax(dim=1)).sum())
        total += 3
    torch.cuda.synchronize()
    kv.free()
    print(f'fp8 decode vs bf16: worst logits rel-L2 {worst:.3e}, greedy picks equal {agree}/{total}')
    assert worst <= 1.5e-1, worst
    assert agree >= total * 0.6, (agree, total)                # flat random-weight logits: most picks still agree
    # with and without the decode layout of the e4m3 copies: the same bits (a second context built under CR_DECODE_LAYOUT=0)
    import os
    fp8_logits, fp8_ids = run(True)
    os.environ['CR_DECODE_LAYOUT'] = '0'
    try:
        plain = Engine(dims, max_pos=1024)
        plain.load_state_dict(sd); plain.load_rope(); plain.finalize()
        plain.enable_fp8_decode(True)
    finally:
        del os.environ['CR_DECODE_LAYOUT']
    kvp = plain.kv_alloc(3, 512)
    plain.prefill_batch(kvp, [0, 1, 2], prompts)
    for t in range(steps):
        assert torch.equal(plain.decode(kvp, [0, 1, 2], want_logits=True).float().cpu(), fp8_logits[t]), t
    assert [kvp.generated(i) for i in range(3)] == fp8_ids
    kvp.free(); plain.close()
    # switching back restores the bf16 results bit for bit
    again, ids2 = run(False)
    assert ids2 == ref_ids and all(torch.equal(a, b) for a, b in zip(again, ref_logits))
    eng.close()
