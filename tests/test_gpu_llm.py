"""InternLM2 prefill + greedy decode parity on a real MI355X vs the CPU oracle (2 layers, full width).

Logit tolerance ("fp16 logit tolerance" of BASELINE.json's north_star), calibrated rather than guessed: on these
very weights/prompts the bf16 CPU oracle differs from an fp32 evaluation of the same network by max 0.085-0.110 and
rel-L2 1.7e-2-2.3e-2 (logits up to |5.5|; measured in the build container, see DESIGN.md "Tolerances").  A second
bf16 implementation with another accumulation order cannot be closer to the oracle than the oracle is to exact
arithmetic, so the bound is that noise floor: |diff| <= 0.12 and rel-L2 <= 2.5e-2.
Token parity: greedy ids must match the oracle exactly, except where the oracle's own top-2 margin is inside that
logit tolerance (a near-tie) — then the test continues teacher-forced and reports the position.
"""
import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu
ATOL = 0.12
RTOL_L2 = 2.5e-2


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope='module')
def setup():
    from callireader_amd.engine import Engine
    dims = ModelDims.reduced(vit_layers=1, llm_layers=2, rs_depth=1, vocab=8201)      # ragged vocab like 92553
    sd = synthetic.make_state_dict(dims, parts=('llm',), seed=0)
    eng = Engine(dims, max_pos=2048)
    eng.load_state_dict(sd)
    eng.load_rope()
    eng.finalize()
    return dict(eng=eng, dims=dims, sd=sd)


def prompt(S, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(1, S, 4096, generator=g) * 0.02).to(torch.bfloat16)


def oracle_run(sd, emb, steps, penalty=1.0):
    from oracle import generate
    with torch.no_grad():
        ids, logits = generate.greedy_generate(sd, 2, emb, max_new_tokens=steps, eos_token_id=-1,
                                               repetition_penalty=penalty, return_logits=True)
    return ids[0].tolist(), logits


def check_stream(eng, kv, seq, emb, ref_ids, ref_logits, penalty=1.0):
    """Teacher-forced walk along the oracle's ids; returns the list of positions where the greedy pick differed."""
    lg = eng.prefill(kv, seq, emb.cuda(), penalty=penalty, want_logits=True)
    torch.cuda.synchronize()
    diverged = []
    for t in range(len(ref_ids)):
        got, ref = lg.float().cpu().reshape(-1), ref_logits[t]
        assert rel_l2(got, ref) <= RTOL_L2, (t, rel_l2(got, ref))
        assert float((got - ref).abs().max()) <= ATOL, (t, float((got - ref).abs().max()))
        picked = kv.generated(seq)[t]
        if picked != ref_ids[t]:
            # only acceptable as a MEASURED near-tie of the (penalised) scores: the oracle's gap between its pick and ours must be
            # covered by the logit differences measured at exactly those two ids (the HIP scores straddle), and stay inside the
            # calibrated tolerance; nothing is excused by a row-wide bound and there is no allowance per stream
            from oracle.generate import near_tie_straddles
            ok, gap, d_ref, d_hip = near_tie_straddles(ref, got, ref_ids[t], picked, ref_ids[:t], penalty, ATOL)
            print(f'  step {t}: pick differs: oracle gap {gap:.4f}, measured d(ref id) {d_ref:+.4f}, d(hip id) {d_hip:+.4f}')
            assert ok, (t, picked, ref_ids[t], gap, d_ref, d_hip)
            diverged.append(t)
        if t + 1 < len(ref_ids):
            lg = eng.decode(kv, [seq], penalty=penalty, force_tokens=torch.tensor([ref_ids[t]]), want_logits=True)
            torch.cuda.synchronize()
    return diverged


@pytest.mark.parametrize('S', [300, 17, 129])
def test_prefill_and_teacher_forced_decode(setup, S):
    eng, sd = setup['eng'], setup['sd']
    emb = prompt(S, 100 + S)
    ref_ids, ref_logits = oracle_run(sd, emb, 6)
    kv = eng.kv_alloc(1, 512)
    div = check_stream(eng, kv, 0, emb, ref_ids, ref_logits)
    assert kv.length(0) == S + 5
    if div:
        print(f'S={S}: greedy picks differed from the oracle at measured near-ties, steps {div}')
    kv.free()


def test_free_running_greedy_tokens(setup):
    eng, sd = setup['eng'], setup['sd']
    emb = prompt(200, 7)
    ref_ids, ref_logits = oracle_run(sd, emb, 12)
    kv = eng.kv_alloc(1, 512)
    eng.prefill(kv, 0, emb.cuda())
    for _ in range(11):
        eng.decode(kv, [0])
    got = kv.generated(0)
    assert len(got) == 12
    first_div = next((i for i, (a, b) in enumerate(zip(got, ref_ids)) if a != b), None)
    if first_div is not None:
        # a free-running stream may leave the oracle's only at a MEASURED near-tie (check_stream's rule, round-3 verdict): walk the
        # oracle's ids teacher-forced up to the step, take the HIP logits there, and require that they straddle the oracle's gap
        from oracle.generate import near_tie_straddles
        kv.reset()
        lg = eng.prefill(kv, 0, emb.cuda(), want_logits=True)
        for t in range(first_div):
            lg = eng.decode(kv, [0], force_tokens=torch.tensor([ref_ids[t]]), want_logits=True)
        torch.cuda.synchronize()
        ok, gap, d_ref, d_hip = near_tie_straddles(ref_logits[first_div], lg.float().cpu().reshape(-1), ref_ids[first_div], got[first_div],
                                                   ref_ids[:first_div], 1.0, ATOL)
        assert ok, f'diverged at {first_div} without a measured near-tie (oracle gap {gap:.4f}, d(ref id) {d_ref:+.4f}, d(hip id) {d_hip:+.4f})'
    else:
        assert got == ref_ids
    kv.free()


def test_repetition_penalty(setup):
    eng, sd = setup['eng'], setup['sd']
    emb = prompt(64, 11)
    ref_ids, ref_logits = oracle_run(sd, emb, 8, penalty=1.5)
    kv = eng.kv_alloc(1, 256)
    div = check_stream(eng, kv, 0, emb, ref_ids, ref_logits, penalty=1.5)
    if div:
        print(f'penalty 1.5: greedy picks differed from the oracle at measured near-ties, steps {div}')
    kv.free()


def test_batched_decode_equals_single(setup):
    """Pages decode together in one batch; every row must equal its own single-sequence run bit for bit."""
    eng = setup['eng']
    embs = [prompt(300, 21), prompt(77, 22), prompt(130, 23)]
    singles = []
    for e in embs:
        kv = eng.kv_alloc(1, 512)
        eng.prefill(kv, 0, e.cuda())
        for _ in range(6):
            eng.decode(kv, [0])
        singles.append(kv.generated(0))
        kv.free()
    kv = eng.kv_alloc(4, 512)
    for i, e in enumerate(embs):
        eng.prefill(kv, i + 1, e.cuda())           # slots 1..3: slot 0 stays empty on purpose
    for _ in range(6):
        eng.decode(kv, [3, 1, 2])                  # arbitrary order
    for i in range(3):
        assert kv.generated(i + 1) == singles[i]
        assert kv.length(i + 1) == embs[i].shape[1] + 6
    assert kv.length(0) == 0
    kv.free()


def test_a_rows_bits_do_not_depend_on_which_kernels_its_batch_takes(setup):
    """One page decoded alone (gemm_decode.hip's fused kernels), inside a 9-row and inside a 16-row batch (gemm_skinny.hip's K-sliced partials
    for wqkv / wo / w2, gemm_stream.hip for w1|w3 and the LM head): the logits of every step and the ids must be the same bits (round-4 advice:
    the 3-row batches of test_batched_decode_equals_single never leave the fused kernels)."""
    eng = setup['eng']
    lens = [300, 77, 130, 17, 64, 255, 256, 31, 129, 40, 200, 90, 5, 150, 33, 61]
    embs = [prompt(S, 700 + i) for i, S in enumerate(lens)]
    kv1 = eng.kv_alloc(1, 512)
    eng.prefill(kv1, 0, embs[0].cuda())
    alone = [eng.decode(kv1, [0], penalty=1.2, want_logits=True).clone() for _ in range(5)]
    ids_alone = kv1.generated(0)
    kv1.free()
    for rows in (9, 16):
        kv = eng.kv_alloc(rows, 512)
        for i in range(rows):
            eng.prefill(kv, i, embs[i].cuda())
        order = list(range(rows - 1, -1, -1))                         # the page is the LAST row of the batch
        for step in range(5):
            lg = eng.decode(kv, order, penalty=1.2, want_logits=True)
            torch.cuda.synchronize()
            assert torch.equal(lg[rows - 1], alone[step][0]), (rows, step, float((lg[rows - 1] - alone[step][0]).abs().max()))
        assert kv.generated(0) == ids_alone
        kv.free()


def test_reloading_a_weight_rebuilds_its_decode_layout(setup):
    """Round-4 advice (medium): the decode-layout copies cr_finalize keeps of every LLM linear must follow a reload.  A finalized context reloads
    ONE wqkv, ONE w2 and the LM head (w1 / w3 are long released, so nothing else says "language model" to cr_finalize), finalizes again, and must
    decode exactly like a fresh context built from the changed state dict -- prefill (nn.Linear layout) and decode (decode layout) alike."""
    from callireader_amd.engine import Engine
    from callireader_amd._binding import CalliReaderError
    dims, sd = setup['dims'], dict(setup['sd'])
    a = Engine(dims, max_pos=2048)
    a.load_state_dict(sd); a.load_rope(); a.finalize()
    emb = prompt(90, 77).cuda()
    kv = a.kv_alloc(1, 256)
    a.prefill(kv, 0, emb)
    before = a.decode(kv, [0], want_logits=True).clone()
    kv.free()
    g = torch.Generator().manual_seed(9)
    changed = {}
    for k in ('language_model.model.layers.0.attention.wqkv.weight', 'language_model.model.layers.1.feed_forward.w2.weight', 'language_model.output.weight'):
        changed[k] = (sd[k].float() + 0.01 * torch.randn(sd[k].shape, generator=g)).to(torch.bfloat16)
    for k, v in changed.items():
        a.load_weight(k, v)
    a.finalize()
    sd.update(changed)
    b = Engine(dims, max_pos=2048)
    b.load_state_dict(sd); b.load_rope(); b.finalize()
    outs = []
    for e in (a, b):
        kv = e.kv_alloc(1, 256)
        first = e.prefill(kv, 0, emb, want_logits=True).clone()
        steps = [e.decode(kv, [0], want_logits=True).clone() for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((first, steps, kv.generated(0)))
        kv.free()
    assert torch.equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert torch.equal(x, y)
    assert outs[0][2] == outs[1][2]
    assert not torch.equal(outs[0][1][0], before)                     # the reload did change the logits
    # one of the w1 / w3 pair alone cannot be re-interleaved (the other original was released): refused, not silently ignored
    a.load_weight('language_model.model.layers.0.feed_forward.w1.weight', setup['sd']['language_model.model.layers.0.feed_forward.w1.weight'])
    with pytest.raises(CalliReaderError):
        a.finalize()
    a.close(); b.close()


def test_reloading_w1_and_w3_with_the_fp8_options_on_rebuilds_their_e4m3_copies(setup):
    """Round-5 advice (medium): the e4m3 copies of w1|w3 are named after the INTERLEAVED tensor (fp8. / fp8s. / fp8dl.derived.w13.N), which cr_load_weight's
    invalidation by the reloaded tensor's own name cannot see.  A context with fp8 decode + fp8 prefill on reloads w1 and w3 of one layer, finalizes, re-enables
    the options and must give a fresh context's logits (prefill: e4m3 x e4m3 w1|w3; decode: e4m3-weight stream) -- and the options are OFF in between."""
    from callireader_amd.engine import Engine
    dims, sd = setup['dims'], dict(setup['sd'])

    def run(e):
        emb = prompt(300, 78).cuda()                                  # >= 256 rows: the prefill takes the e4m3 matrix-core linears
        kv = e.kv_alloc(1, 512)
        first = e.prefill(kv, 0, emb, want_logits=True).clone()
        steps = [e.decode(kv, [0], want_logits=True).clone() for _ in range(3)]
        torch.cuda.synchronize()
        kv.free()
        return [first] + steps
    a = Engine(dims, max_pos=2048)
    a.load_state_dict(sd); a.load_rope(); a.finalize()
    a.enable_fp8_mfma(True, level=2); a.enable_fp8_decode(True)
    before = run(a)
    g = torch.Generator().manual_seed(10)
    changed = {}
    for k in ('language_model.model.layers.0.feed_forward.w1.weight', 'language_model.model.layers.0.feed_forward.w3.weight'):
        changed[k] = (sd[k].float() + 0.02 * torch.randn(sd[k].shape, generator=g)).to(torch.bfloat16)
        a.load_weight(k, changed[k])
    a.finalize()
    bf16_after = run(a)                                               # the options went off with the copies they read
    a.enable_fp8_mfma(True, level=2); a.enable_fp8_decode(True)
    after = run(a)
    sd.update(changed)
    b = Engine(dims, max_pos=2048)
    b.load_state_dict(sd); b.load_rope(); b.finalize()
    plain = run(b)
    b.enable_fp8_mfma(True, level=2); b.enable_fp8_decode(True)
    fresh = run(b)
    for x, y in zip(after, fresh):
        assert torch.equal(x, y)
    for x, y in zip(bf16_after, plain):
        assert torch.equal(x, y)
    assert not torch.equal(after[0], before[0]) and not torch.equal(after[1], before[1])
    a.close(); b.close()


@pytest.mark.parametrize('rows', [1, 2, 3, 4, 5, 7, 8, 9])
def test_fused_small_batch_decode_gives_the_separate_kernels_bits(setup, rows):
    """Batches of <= 8 rows decode through gemm_decode.hip (RMSNorm prologues, RoPE + cache-write epilogue, residual-add epilogues: six
    launches per layer); CR_DECODE_FUSED=0 keeps the separate kernels.  Logits, cache rows and ids must be the SAME BITS, step after step:
    a row's result may not depend on which path its batch size selects (a 9-row batch takes the separate kernels in both engines;
    5..8 rows take the two-tiles-per-workgroup forms of w1|w3 and the LM head)."""
    import os
    from callireader_amd.engine import Engine
    eng = setup['eng']
    os.environ['CR_DECODE_FUSED'] = '0'
    try:
        ref = Engine(setup['dims'], max_pos=2048)
    finally:
        del os.environ['CR_DECODE_FUSED']
    ref.share_weights_from(eng)
    lens = [300, 77, 130, 17, 64, 255, 256, 31, 129, 40, 200, 90, 5, 150, 33, 61][:rows]
    embs = [prompt(S, 500 + i) for i, S in enumerate(lens)]
    kva, kvb = eng.kv_alloc(rows + 1, 512), ref.kv_alloc(rows + 1, 512)
    for i, e in enumerate(embs):
        eng.prefill(kva, i + 1, e.cuda())
        ref.prefill(kvb, i + 1, e.cuda())
    order = list(range(rows, 0, -1))                                  # slot 0 stays empty, rows in reverse order
    for step in range(5):
        la = eng.decode(kva, order, penalty=1.3, want_logits=True).clone()
        lb = ref.decode(kvb, order, penalty=1.3, want_logits=True).clone()
        torch.cuda.synchronize()
        assert torch.equal(la, lb), (rows, step, float((la - lb).abs().max()))
    for i in range(rows):
        assert kva.generated(i + 1) == kvb.generated(i + 1)
        pos = lens[i] + 3
        for layer in (0, 1):
            for which in (0, 1):
                assert torch.equal(kva.read(layer, i + 1, pos, which), kvb.read(layer, i + 1, pos, which)), (i, layer, which)
    kva.free(); kvb.free()
    ref.close()


@pytest.mark.parametrize('rows', [1, 5, 8])
def test_decode_layout_of_the_weights_gives_the_same_bits(rows):
    """cr_finalize keeps a second copy of every LLM linear in the decode layout (one contiguous KiB per 16-row tile and 32-deep k-step, wqkv in
    its RoPE tile order; cr_op_decode_swizzle) and the small-batch decode GEMMs stream that copy: the same values reach the same registers, so
    every output must be the same bits as from the nn.Linear layout -- all five GEMMs, a vocabulary that is not a multiple of 16."""
    from callireader_amd import engine as E
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(rows)
    D, FF, V, QKV = 4096, 14336, 8201, 6144
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()
    x, ao, act = rnd(rows, D), rnd(rows, D, sc=0.5), rnd(rows, FF, sc=0.5)
    gm = (1 + 0.1 * torch.randn(D, device=dev, generator=g)).bfloat16()
    W = {0: rnd(QKV, D, sc=0.02), 1: rnd(D, D, sc=0.02), 2: rnd(2 * FF, D, sc=0.02), 3: rnd(D, FF, sc=0.02), 4: rnd(V, D, sc=0.02)}
    S = {k: E.op_decode_swizzle(k, w) for k, w in W.items()}
    assert S[4].numel() == ((V + 15) // 16) * 16 * D
    max_tokens = 64

    def rope():
        return dict(cos=rope_c, sin=rope_s, q_out=torch.zeros(rows, D, device=dev, dtype=torch.bfloat16),
                    kc=torch.zeros(rows, 8, max_tokens, 128, device=dev, dtype=torch.bfloat16), vc=torch.zeros(rows, 8, max_tokens, 128, device=dev, dtype=torch.bfloat16),
                    seqs=torch.arange(rows, device=dev, dtype=torch.int32), lens=torch.full((rows,), 9, device=dev, dtype=torch.int32), max_tokens=max_tokens)
    rope_c, rope_s = rnd(max_tokens, 128), rnd(max_tokens, 128)
    outs = []
    for sw in (False, True):
        pick = lambda k: S[k] if sw else None
        r = rope()
        E.op_decode_gemm(0, W[0], rows, xres=x, gamma=gm, rope=r, swizzled=pick(0))
        xa = x.clone()
        E.op_decode_gemm(1, W[1], rows, X=ao, xio=xa, swizzled=pick(1))
        a13 = torch.zeros(rows, FF, device=dev, dtype=torch.bfloat16)
        E.op_decode_gemm(2, W[2], rows, xres=x, gamma=gm, C_out=a13, swizzled=pick(2))
        xb = x.clone()
        E.op_decode_gemm(3, W[3], rows, X=act, xio=xb, swizzled=pick(3))
        lg = torch.zeros(rows, V, device=dev, dtype=torch.float32)
        E.op_decode_gemm(4, W[4], rows, xres=x, gamma=gm, C_out=lg, swizzled=pick(4))
        torch.cuda.synchronize()
        outs.append((r['q_out'], r['kc'], r['vc'], xa, a13, xb, lg))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][6].abs().max()) > 0 and float(outs[0][1].abs().max()) > 0


def test_decode_graph_replay_equals_plain_launches(setup):
    """CR_DECODE_GRAPH=1 (opt-in): the decode step captured as a hipGraph and replayed -- across a change of the number
    of attention splits (a second capture) and with prefills in between -- generates exactly the plain path's ids."""
    import os
    from callireader_amd.engine import Engine
    embs = [prompt(250, 41), prompt(120, 42), prompt(33, 43)]

    def run(eng):
        kv = eng.kv_alloc(3, 512)
        for i, e in enumerate(embs):
            eng.prefill(kv, i, e.cuda())
        for _ in range(12):                          # sequence 0 crosses 256 cached tokens: the split count grows
            eng.decode(kv, [0, 1, 2])
        out = [kv.generated(i) for i in range(3)]
        kv.free()
        return out
    plain = run(setup['eng'])
    os.environ['CR_DECODE_GRAPH'] = '1'
    try:
        eng2 = Engine(setup['dims'], max_pos=2048)
    finally:
        del os.environ['CR_DECODE_GRAPH']
    eng2.load_state_dict(setup['sd'])
    eng2.load_rope()
    eng2.finalize()
    assert run(eng2) == plain
    assert run(eng2) == plain                        # replays of the cached graphs on a fresh cache


def test_prefill_batch_equals_single(setup):
    """Prompts of several pages share the linear layers' GEMMs; each page must get exactly its own single-prefill result
    (the tiled GEMM kernels accumulate K in the same order for every row, attention stays per page)."""
    eng = setup['eng']
    embs = [prompt(300, 31), prompt(77, 32), prompt(130, 33)]
    single_logits, single_ids = [], []
    for e in embs:
        kv = eng.kv_alloc(1, 512)
        single_logits.append(eng.prefill(kv, 0, e.cuda(), want_logits=True).clone())
        for _ in range(3):
            eng.decode(kv, [0])
        single_ids.append(kv.generated(0))
        kv.free()
    kv = eng.kv_alloc(4, 512)
    lg = eng.prefill_batch(kv, [2, 0, 3], [e.cuda() for e in embs], want_logits=True)
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(lg[i], single_logits[i])
    for _ in range(3):
        eng.decode(kv, [0, 2, 3])
    assert [kv.generated(s) for s in (2, 0, 3)] == single_ids
    assert [kv.length(s) for s in (2, 0, 3)] == [303, 80, 133] and kv.length(1) == 0
    from callireader_amd._binding import CalliReaderError
    with pytest.raises(CalliReaderError):
        eng.prefill_batch(kv, [1, 1], [embs[1].cuda(), embs[1].cuda()])
    kv.free()


def test_final_layer_on_the_last_rows_only_gives_the_same_bits(setup):
    """Prefill, final decoder layer: attention / wo / RMSNorm / w1|w3 / w2 run on each page's LAST row only (round-4 verdict, item 3a: nobody reads the
    other rows; the reference computes them, modeling_internlm2.py:1081-1082); K / V of every row still reach the cache.  CR_PREFILL_LAST_ROWS=0
    keeps every row.  Logits, the ids of the following decode steps and the final layer's cache rows must be the SAME BITS -- page lengths on both sides
    of the attention kernel's 32-row waves and 128-row blocks, a batch large enough for the 256x256 GEMM kernel (>= 2048 rows) and single short prompts."""
    import os
    from callireader_amd.engine import Engine
    eng = setup['eng']
    os.environ['CR_PREFILL_LAST_ROWS'] = '0'
    try:
        ref = Engine(setup['dims'], max_pos=2048)
    finally:
        del os.environ['CR_PREFILL_LAST_ROWS']
    ref.share_weights_from(eng)
    lens = [300, 77, 130, 1500, 33, 32, 64, 65, 1, 129, 128, 97]
    embs = [prompt(S, 900 + i).cuda() for i, S in enumerate(lens)]
    assert sum(lens) >= 2048
    outs = []
    for e in (eng, ref):
        kv = e.kv_alloc(len(lens) + 1, 1600)
        lg_batch = e.prefill_batch(kv, list(range(1, len(lens) + 1)), embs, penalty=1.1, want_logits=True).clone()
        singles = []
        kv1 = e.kv_alloc(1, 1600)
        for x in (embs[4], embs[8], embs[0]):
            kv1.reset()
            singles.append(e.prefill(kv1, 0, x, want_logits=True).clone())
        kv1.free()
        steps = [e.decode(kv, list(range(1, len(lens) + 1)), penalty=1.1, want_logits=True).clone() for _ in range(3)]
        torch.cuda.synchronize()
        cache = [kv.read(1, i + 1, lens[i] - 1, which) for i in (0, 3, 8) for which in (0, 1)]        # final layer (of 2), last prompt position
        outs.append((lg_batch, singles, steps, [kv.generated(i + 1) for i in range(len(lens))], cache))
        kv.free()
    a, b = outs
    assert torch.equal(a[0], b[0]), float((a[0] - b[0]).abs().max())
    for x, y in zip(a[1] + a[2] + a[4], b[1] + b[2] + b[4]):
        assert torch.equal(x, y)
    assert a[3] == b[3]
    # and a page prefilled in the batch equals the same page prefilled alone (both with the shortcut)
    assert torch.equal(a[0][4], a[1][0]) and torch.equal(a[0][8], a[1][1]) and torch.equal(a[0][0], a[1][2])
    ref.close()


def test_kv_reset_and_limits(setup):
    from callireader_amd._binding import CalliReaderError
    eng = setup['eng']
    kv = eng.kv_alloc(1, 64)
    with pytest.raises(CalliReaderError):
        eng.prefill(kv, 0, prompt(65, 1).cuda())           # does not fit
    eng.prefill(kv, 0, prompt(60, 1).cuda())
    a = kv.generated(0)
    kv.reset(0)
    assert kv.length(0) == 0 and kv.generated(0) == []
    eng.prefill(kv, 0, prompt(60, 1).cuda())
    assert kv.generated(0) == a
    with pytest.raises(CalliReaderError):
        eng.decode(kv, [0, 0])
    kv.free()


def test_embed_splice(setup):
    from oracle import generate
    from callireader_amd._binding import CalliReaderError
    eng, sd = setup['eng'], setup['sd']
    g = torch.Generator().manual_seed(3)
    IMG, REF = 8000, 7999
    ids = torch.randint(0, 7000, (1, 1500), generator=g)
    ids[0, 10:10 + 512] = IMG
    ids[0, 1200:1209] = REF
    ids[0, 1300:1303] = REF
    vit = torch.randn(2, 256, 4096, generator=g).bfloat16()
    ref = torch.randn(12, 4096, generator=g).bfloat16()
    exp = generate.splice_embeddings(sd, ids, vit, ref, IMG, REF)[0]
    got = eng.embed_splice(ids, vit.cuda(), ref.cuda(), img_id=IMG, ref_id=REF)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), exp)
    got = eng.embed_splice(ids, vit.cuda(), None, img_id=IMG, ref_id=REF)
    assert torch.equal(got.cpu(), generate.splice_embeddings(sd, ids, vit, None, IMG, REF)[0])
    got = eng.embed_splice(ids)
    assert torch.equal(got.cpu(), generate.splice_embeddings(sd, ids)[0])
    with pytest.raises(CalliReaderError):
        eng.embed_splice(ids, vit[:1].cuda(), ref.cuda(), img_id=IMG, ref_id=REF)     # count mismatch
    with pytest.raises(CalliReaderError):
        eng.embed_splice(torch.zeros(1, 8, dtype=torch.long), vit.cuda(), None, img_id=IMG)   # no <IMG_CONTEXT>
