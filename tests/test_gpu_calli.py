"""CalliAlign parity on a real MI355X: resampler, cosine VQ, de-normalisation vs the CPU oracle."""

import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope='module')
def setup():
    from callireader_amd.engine import Engine
    from oracle import calli_align
    dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=2, vocab=5003)     # ragged vocab
    sd = synthetic.make_state_dict(dims, parts=('resampler', 'vq'), seed=0)
    eng = Engine(dims)
    eng.load_state_dict(sd)
    eng.finalize()
    g = torch.Generator().manual_seed(5)
    feats = (torch.randn(5, 256, 4096, generator=g) * 0.7).to(torch.bfloat16)
    with torch.no_grad():
        rs = calli_align.resampler_forward(sd, feats, dims.rs_depth)
    return dict(eng=eng, dims=dims, sd=sd, feats=feats, rs=rs, oracle=calli_align)


def test_resampler(setup):
    out = setup['eng'].resample(setup['feats'].cuda())
    torch.cuda.synchronize()
    got, ref = out.float().cpu(), setup['rs'].float()
    assert got.shape == ref.shape == (5, 3, 4096)
    assert rel_l2(got, ref) <= 1.5e-2
    assert float((got - ref).abs().max()) <= 8e-2 * float(ref.abs().max())


def test_resampler_single_tile(setup):
    out = setup['eng'].resample(setup['feats'][:1].cuda())
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), setup['rs'][:1].float()) <= 1.5e-2


def test_perceiver_attention_without_scratch_gives_round_1s_bits(setup):
    """Round-5 verdict, item 6: the perceiver attention kernel was rewritten (a real loop over the dims with the next chunk in flight, the V block staged through LDS
    in one burst: no scratch, 178 registers).  Every fp32 sum keeps the order of round 1's kernel, which stays selectable (CR_PERCEIVER_ATTN_V1=1 when the context
    is created): the two contexts share the weights and must produce the same bits, on a batch that spans two resampler chunks (RS_CHUNK = 252)."""
    import os
    from callireader_amd.engine import Engine
    eng = setup['eng']
    os.environ['CR_PERCEIVER_ATTN_V1'] = '1'
    try:
        old = Engine(setup['dims'])
    finally:
        del os.environ['CR_PERCEIVER_ATTN_V1']
    old.share_weights_from(eng)
    g = torch.Generator().manual_seed(23)
    feats = (torch.randn(260, 256, 4096, generator=g) * 0.7).to(torch.bfloat16).cuda()
    a, b = eng.resample(feats), old.resample(feats)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.equal(eng.resample(setup['feats'].cuda()), old.resample(setup['feats'].cuda()))
    old.close()


def test_tile_results_do_not_depend_on_the_batch(setup):
    """A tile's pseudo tokens must be the same whether it is resampled alone, in a rank's shard of 7, or with every other tile of
    a batch (the 8-rank flow of scripts/dist_check.py shards 55 tiles as 7 + ... + 6): bit for bit, through resampler, VQ and
    de-normalisation.  30 tiles = 90 latent rows crosses the 64-row line where the GEMM dispatcher changes kernel class."""
    eng = setup['eng']
    g = torch.Generator().manual_seed(17)
    feats = (torch.randn(30, 256, 4096, generator=g) * 0.7).to(torch.bfloat16).cuda()
    whole = eng.resample(feats)
    for step in (1, 7, 21, 22):
        parts = torch.cat([eng.resample(feats[i:i + step]) for i in range(0, 30, step)])
        assert torch.equal(parts, whole), f'resampler output depends on the batch (chunks of {step})'
    idx = eng.vq(whole)
    assert torch.equal(torch.cat([eng.vq(whole[i:i + 7]) for i in range(0, 30, 7)]), idx)
    back = eng.denorm(whole, idx)
    assert torch.equal(torch.cat([eng.denorm(whole[i:i + 7], idx[i:i + 7]) for i in range(0, 30, 7)]), back)


def hip_similarities(E, x_row_normalised, table_normalised, i_ref, i_hip):
    """The HIP tiled GEMM's own similarities of one normalised query row against the reference's and the HIP path's table rows: (at i_ref, at i_hip)."""
    tn = table_normalised
    rows = torch.stack([tn[i_ref], tn[i_hip]] + [tn[(i_ref + 1 + k) % tn.shape[0]] for k in range(14)])     # 16 table rows, the two candidates first
    s = E.op_gemm(0, x_row_normalised[None].expand(16, -1).contiguous().cuda(), rows.cuda(), kernel=1).float().cpu()[0]
    return float(s[0]), float(s[1])


def test_vq_planted_and_random(setup):
    o, sd = setup['oracle'], setup['sd']
    table = sd['normed_emb.weight']
    q = setup['rs'].clone()
    q[0, 0] = table[123] * 3.0
    q[1, 2] = table[5002] * 0.5          # last row of a ragged table
    q[2, 1] = table[0] * 2.0
    ridx, rcos = o.vq_cos_sim(table, q, use_dynamic_p=True)
    idx, cos = setup['eng'].vq(q.cuda(), with_cos=True)
    torch.cuda.synchronize()
    idx, cos = idx.cpu(), cos.float().cpu()
    assert idx.shape == (5, 3) and idx.dtype == torch.int64
    assert int(idx[0, 0]) == 123 and int(idx[1, 2]) == 5002 and int(idx[2, 1]) == 0
    # cosine of the planted rows is ~1 on both sides; elsewhere the max cosine must agree to bf16 resolution
    assert torch.allclose(cos, rcos.float(), atol=8e-3)
    # Index work is exact work (round-3 verdict): an index may differ from the oracle's only at a MEASURED tie -- oracle/calli_align.py: vq_tie_rule, the ONE
    # rule shared with tests/test_gpu_full_depth.py and scripts/real_checkpoint_parity.py.  sim = the oracle's own bf16 similarity matrix (similarity.py:17-19).
    # Where the oracle's top-2 gap exceeds 2^-7 the indices must be equal outright; the HIP similarities at the two rows come from the tiled kernel on the
    # same normalised operands.
    from callireader_amd import engine as E
    xn = torch.nn.functional.normalize(q, p=2, dim=2)
    tn = torch.nn.functional.normalize(table, p=2, dim=1)
    sim = torch.matmul(xn, tn.t()).float()
    top2 = sim.topk(2, dim=2).values
    flat_x = xn.reshape(-1, xn.shape[-1])
    n_diff = 0
    for b in range(idx.shape[0]):
        for j in range(idx.shape[1]):
            i_h, i_o = int(idx[b, j]), int(ridx[b, j])
            if i_h == i_o:
                continue
            n_diff += 1
            assert float(top2[b, j, 0] - top2[b, j, 1]) <= 2.0 ** -7, (b, j, 'the oracle has a clear maximum here')
            s_hip = hip_similarities(E, flat_x[b * idx.shape[1] + j], tn, i_o, i_h)
            ok, gap, step = o.vq_tie_rule(sim[b, j, i_o], sim[b, j, i_h], s_hip[0], s_hip[1])
            print(f'  VQ ({b},{j}): HIP {i_h} vs oracle {i_o}: oracle gap {gap:.3e} (one step {step:.3e}); HIP similarities {s_hip[1]:.6f} / {s_hip[0]:.6f}')
            assert ok, (b, j, i_h, i_o, gap, step, s_hip)
    print(f'VQ: {n_diff} of {idx.numel()} indices differ from the oracle, all at measured ties')


def test_denorm_branches(setup):
    o, sd, eng = setup['oracle'], setup['sd'], setup['eng']
    table, mu, sigma = sd['normed_emb.weight'], sd['calli.mu'], sd['calli.sigma']
    x = setup['rs']
    idx = torch.tensor([[0, 3, 5], [7, 0, 2], [9, 9, 0], [1, 2, 3], [0, 0, 4]])
    cos = torch.tensor([[0.9, 0.4, 0.5], [0.6, 0.1, 0.95], [0.7, 0.2, 0.3], [0.99, 0.5, 0.51], [0.0, 1.0, 0.49]]).bfloat16()
    for drop_zero in (False, True):
        for hard in (False, True):
            ref, _ = o.denormalise(x, idx, table, mu, sigma, drop_zero=drop_zero, hard_vq=hard, cos=cos)
            got = eng.denorm(x.cuda(), idx.cuda(), cos.cuda(), drop_zero=drop_zero, hard_vq=hard)
            torch.cuda.synchronize()
            assert got.shape == ref.shape
            assert torch.equal(got.cpu(), ref), (drop_zero, hard)      # elementwise bf16 ops: bit-exact


def test_denorm_fp32_params(setup):
    from callireader_amd.engine import Engine
    sd = dict(setup['sd'])
    eng = Engine(setup['dims'])
    eng.load_weight('normed_emb.weight', sd['normed_emb.weight'])
    eng.load_weight('calli.mu', sd['calli.mu'].float())
    eng.load_weight('calli.sigma', sd['calli.sigma'].float())
    eng.finalize()
    x = setup['rs']
    idx = torch.randint(0, 5003, (5, 3))
    got = eng.denorm(x.cuda(), idx.cuda())
    torch.cuda.synchronize()
    flat = x.reshape(-1, 4096)
    ref = (flat * sd['calli.sigma'].float()[idx.reshape(-1)] + sd['calli.mu'].float()[idx.reshape(-1)]).to(torch.bfloat16)
    assert torch.equal(got.cpu(), ref)
