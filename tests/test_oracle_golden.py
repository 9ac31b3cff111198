"""Pin the CPU oracle to vectors produced by the reference's own modules.

tests/golden/reference_vectors.npz is written by scripts/make_golden.py, which
imports /root/reference in the build container.  Nothing here reads
/root/reference.  The oracle runs the same eager CPU ops as the reference, so
agreement is expected to be (near) bit-exact; tolerances below are 1 bf16 ulp.
"""
import os
import json

import numpy as np
import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from oracle import vision, calli_align, internlm2, generate

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'reference_vectors.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


@pytest.fixture(scope='module')
def dims():
    return ModelDims.reduced(vit_layers=2, llm_layers=2, rs_depth=2)


@pytest.fixture(scope='module')
def vis(dims):
    sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1'), seed=0)
    px = synthetic.make_pixels(2, seed=1)
    with torch.no_grad():
        emb = vision.vit_embeddings(sd, px)
        lay0 = vision.vit_layer(sd, 0, emb)
        last = vision.vit_forward(sd, px, dims.vit_layers)
        feat = vision.project(sd, last)
    return dict(emb=emb, lay0=lay0, last=last, feat=feat)


def check_sample(gold, prefix, t, atol, rtol=0.0):
    f = t.detach().float().reshape(-1)
    assert f.numel() == int(gold[prefix + '.numel'])
    step = int(gold[prefix + '.step'])
    got = f[::step][:4096].numpy()
    exp = gold[prefix + '.sample']
    np.testing.assert_allclose(got, exp, atol=atol, rtol=rtol)
    assert abs(f.double().sum().item() - float(gold[prefix + '.sum'])) <= 1e-3 * float(gold[prefix + '.abssum']) + 1e-6


def test_meta_matches(dims):
    meta = json.load(open(GOLD.replace('.npz', '.json')))
    assert meta['dims'] == dims.asdict()


def test_vit_embeddings(gold, vis):
    check_sample(gold, 'vit_embeddings', vis['emb'], atol=0.0)


def test_vit_layer(gold, vis):
    check_sample(gold, 'vit_layer0', vis['lay0'], atol=0.0)


def test_vit_last_hidden(gold, vis):
    check_sample(gold, 'vit_last', vis['last'], atol=0.0)


def test_extract_feature(gold, vis):
    check_sample(gold, 'extract_feature', vis['feat'], atol=0.0)


def test_pixel_shuffle_pattern(gold):
    pat = torch.arange(2 * 4 * 4 * 8, dtype=torch.float32).reshape(2, 4, 4, 8)
    np.testing.assert_array_equal(vision.pixel_shuffle(pat, 0.5).numpy(), gold['pixel_shuffle.pattern_out'])


@pytest.fixture(scope='module')
def rs_out(dims, vis):
    sd = synthetic.make_state_dict(dims, parts=('resampler',), seed=0)
    with torch.no_grad():
        return calli_align.resampler_forward(sd, vis['feat'], dims.rs_depth)


def test_resampler(gold, rs_out):
    np.testing.assert_allclose(rs_out.float().numpy(), gold['resampler.full'], atol=0.0)


def test_vq(gold, rs_out):
    vsd = synthetic.make_state_dict(ModelDims.reduced(vocab=4096), parts=('vq',), seed=0)
    table = vsd['normed_emb.weight']
    q = rs_out.clone()
    q[0, 0] = table[123] * 3.0
    q[1, 2] = table[4000] * 0.5
    idx, cos = calli_align.vq_cos_sim(table, q, use_dynamic_p=True)
    np.testing.assert_array_equal(idx.numpy(), gold['vq.indices'])
    np.testing.assert_allclose(cos.float().numpy(), gold['vq.cos'], atol=0.0)
    assert int(idx[0, 0]) == 123 and int(idx[1, 2]) == 4000
    idx1 = calli_align.vq_cos_sim(table, q[:1])
    assert idx1.shape == (3,)
    np.testing.assert_array_equal(idx1.numpy(), gold['vq.indices_T1'])


def test_denormalise_pinned_to_the_reference_tail():
    """oracle.denormalise against what the reference's OWN statements (modeling_internvl_chat.py:602-640, compiled out of
    calli_align at generation time by scripts/make_golden_tail.py) produced: plain, drop_zero, hard VQ (incl. the
    cos == 0.5 boundary), both, a single tile; mu/sigma in fp32 and in bf16 (result dtype follows the promotion)."""
    import tail_cases as mgt
    gold = np.load(os.path.join(os.path.dirname(GOLD), 'tail_vectors.npz'))
    for name, seed, tiles, vocab, pdt, drop_zero, hard_vq in mgt.CASES:
        table, mu, sigma, x, idx, cos = mgt.make_case(seed, tiles, vocab, pdt, drop_zero, hard_vq)
        out, indices = calli_align.denormalise(x, idx, table, mu, sigma, drop_zero=drop_zero, hard_vq=hard_vq, cos=cos if hard_vq else None)
        assert str(out.dtype) == bytes(gold[f'{name}.out_dtype']).decode(), name
        np.testing.assert_array_equal(out.float().numpy(), gold[f'{name}.out'], err_msg=name)
        np.testing.assert_array_equal(indices.numpy(), gold[f'{name}.indices'], err_msg=name)


def test_denormalise_branches():
    # No importable reference function exists for the calli_align tail (it is inline at
    # modeling_internvl_chat.py:602-640), so this checks the restatement against the
    # formula written out by hand on a tiny case, incl. drop_zero and hard-VQ branches.
    torch.manual_seed(0)
    V, D = 16, 8
    table = torch.randn(V, D).bfloat16()
    mu = torch.randn(V, 1).bfloat16()
    sigma = torch.rand(V, 1).bfloat16()
    x = torch.randn(2, 3, D).bfloat16()
    idx = torch.tensor([[0, 3, 5], [7, 0, 2]])
    cos = torch.tensor([[0.9, 0.4, 0.5], [0.6, 0.1, 0.95]]).bfloat16()
    out, _ = calli_align.denormalise(x, idx, table, mu, sigma)
    exp = x.reshape(-1, D) * sigma[idx.reshape(-1)] + mu[idx.reshape(-1)]
    assert torch.equal(out, exp)
    out, _ = calli_align.denormalise(x, idx, table, mu, sigma, drop_zero=True)
    keep = idx.reshape(-1) != 0
    assert out.shape[0] == 4 and torch.equal(out, exp[keep])
    out, _ = calli_align.denormalise(x, idx, table, mu, sigma, hard_vq=True, cos=cos)
    below = (cos <= 0.5).bfloat16().unsqueeze(-1)
    xb = x * (1 - below) + table[idx] * below
    assert torch.equal(out, xb.reshape(-1, D) * sigma[idx.reshape(-1)] + mu[idx.reshape(-1)])
    assert torch.equal(out[1], (table[3] * sigma[3] + mu[3]))


def test_rope_table(gold):
    cos, sin = internlm2.rope_tables(128)
    rows = gold['rope.rows'].tolist()
    np.testing.assert_array_equal(cos[rows].bfloat16().float().numpy(), gold['rope.cos'])
    np.testing.assert_array_equal(sin[rows].bfloat16().float().numpy(), gold['rope.sin'])


@pytest.mark.parametrize('S', [17, 300])
def test_internlm2_prefill_and_greedy(gold, S):
    ldims = ModelDims.reduced(llm_layers=2, vocab=8192)
    sd = synthetic.make_state_dict(ldims, parts=('llm',), seed=0)
    g = torch.Generator().manual_seed(100 + S)
    emb = (torch.randn(1, S, 4096, generator=g) * 0.02).to(torch.bfloat16)
    with torch.no_grad():
        logits, past = internlm2.model_forward(sd, 2, inputs_embeds=emb)
        np.testing.assert_allclose(logits[0, -1].numpy(), gold[f'llm.S{S}.last_logits'], atol=0.0)
        np.testing.assert_allclose(past[0][0][0, :, -1, :].float().numpy(), gold[f'llm.S{S}.k0_last'], atol=0.0)
        np.testing.assert_allclose(past[1][1][0, :, 0, :].float().numpy(), gold[f'llm.S{S}.v1_first'], atol=0.0)
        ids, step_logits = generate.greedy_generate(sd, 2, emb, max_new_tokens=9, eos_token_id=-1, return_logits=True)
    np.testing.assert_array_equal(ids[0, :8].numpy(), gold[f'llm.S{S}.greedy_tokens'])
    got = np.stack([r.numpy() for r in step_logits[1:9]])[:, ::64]
    np.testing.assert_allclose(got, gold[f'llm.S{S}.step_logits_sample'], atol=0.0)


def test_repetition_penalty_semantics():
    s = torch.tensor([2.0, -1.0, 0.5, 3.0])
    out = generate.apply_repetition_penalty(s, [0, 1, 1], 2.0)
    assert out.tolist() == [1.0, -2.0, 0.5, 3.0]
    assert generate.apply_repetition_penalty(s, [0], 1.0) is s


def test_splice_asserts_and_values():
    sd = {'language_model.model.tok_embeddings.weight': torch.arange(20 * 4, dtype=torch.float32).reshape(20, 4)}
    ids = torch.tensor([[1, 7, 7, 9, 2]])
    vit = torch.full((2, 4), -1.0)
    ref = torch.full((1, 4), -2.0)
    out = generate.splice_embeddings(sd, ids, vit, ref, img_context_token_id=7, aligned_token_id=9)
    assert out[0, 1].tolist() == [-1.0] * 4 and out[0, 3].tolist() == [-2.0] * 4
    assert out[0, 0].tolist() == [4.0, 5.0, 6.0, 7.0]
    with pytest.raises(AssertionError):
        generate.splice_embeddings(sd, torch.tensor([[1, 2]]), vit, None, img_context_token_id=7)


def test_repetition_penalty_rule_matches_installed_transformers():
    """The greedy loop as a whole stays unpinned (the reference's transformers 4.45.2 is not installed and the
    installed release cannot drive the reference model), but the one piece of arithmetic in it can be pinned against
    the library's own RepetitionPenaltyLogitsProcessor, whose rule has not changed since 4.45.2: every id already
    generated (duplicates included) sees score/penalty if score > 0 else score*penalty, applied to the ORIGINAL score."""
    lp = pytest.importorskip('transformers.generation.logits_process')
    from oracle.generate import apply_repetition_penalty
    g = torch.Generator().manual_seed(5)
    for penalty in (1.0, 1.2, 2.5):
        scores = torch.randn(257, generator=g) * 3
        scores[7] = 0.0
        generated = [3, 7, 7, 200, 3, 256, 0]
        want = lp.RepetitionPenaltyLogitsProcessor(penalty=penalty)(torch.tensor([generated]), scores[None].clone())[0] \
            if penalty != 1.0 else scores.clone()
        got = apply_repetition_penalty(scores.clone(), generated, penalty)
        assert torch.equal(got, want)


def test_outlier_checkpoints_are_the_same_function_bit_for_bit():
    """callireader_amd/synthetic.py: outlier_transform -- power-of-two re-scaling of matched norm-gain / weight-column pairs (and w3 rows / w2 columns).  It
    commutes with every rounding of the reference's bf16 arithmetic, so the oracle (pinned to the reference at atol 0 above) must return the SAME BITS on the
    plain and on the outlier checkpoints: that is what makes tests/golden/full_depth.npz the golden of every variant (scripts/fp8_schemes.py)."""
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    from oracle import internlm2, vision
    dims = ModelDims.reduced(vit_layers=1, llm_layers=2, rs_depth=1, vocab=2048)
    emb = (torch.randn(1, 24, 4096, generator=torch.Generator().manual_seed(3)) * 0.02).to(torch.bfloat16)
    px = synthetic.make_pixels(1, seed=1)
    outs = []
    for shift in (0, 5, 10):
        sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1', 'llm'), seed=0, outlier_shift=shift)
        with torch.no_grad():
            outs.append((internlm2.model_forward(sd, 2, inputs_embeds=emb, all_logits=False)[0], vision.extract_feature(sd, px, 1)))
        if shift:
            k = 'language_model.model.layers.0.attention_norm.weight'
            plain = synthetic.make_state_dict(dims, parts=('llm',), seed=0)[k]
            assert float((sd[k].float() / plain.float()).max()) == 2.0 ** shift           # the outliers are really there
    for lg, ft in outs[1:]:
        assert torch.equal(lg, outs[0][0]) and torch.equal(ft, outs[0][1])
