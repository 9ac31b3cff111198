"""CPU checks of tests/golden/peaked_streams.npz (scripts/make_golden_peaked.py: free-running greedy streams of the REFERENCE's
InternLM2ForCausalLM, 32 layers, vocabulary 92 553, on the margin-controlled checkpoint of callireader_amd.synthetic).

 * the reference generated exactly the walks the checkpoint was built to produce (so the GPU test's expectation is not an
   accident of one run), stopped on EOS / on the length where it should, and its top-2 margins are what the token-exactness
   test needs: >= 1.0 on >= 90 % of the steps;
 * the oracle's restatement of the transformers 4.45.2 loop pieces (oracle/generate.py: repetition penalty rule, arg-max,
   EOS / length stop) reproduces the picks of the installed release's own RepetitionPenaltyLogitsProcessor /
   EosTokenCriteria / MaxLengthCriteria, which drove the golden run, from the raw logits stored per step.
"""
import json
import os

import numpy as np
import torch

from callireader_amd import synthetic
from callireader_amd.config import ModelDims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'peaked_streams.npz')


def load():
    g = np.load(GOLD)
    return g, json.loads(bytes(g['meta']).decode())


def test_reference_streams_are_the_built_walks():
    g, meta = load()
    plan = synthetic.peaked_plan(ModelDims.full().vocab, meta['start_a'], meta['seed'])
    assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in synthetic.PEAKED.items()} == meta['peaked'], \
        'synthetic.PEAKED changed: regenerate tests/golden/peaked_streams.npz (scripts/make_golden_peaked.py)'
    a, b15, b10 = g['A.ids'].tolist(), g['B15.ids'].tolist(), g['B10.ids'].tolist()
    assert a == plan['chain_a'] == g['built.chain_a'].tolist()
    assert b15 == plan['chain_b'] == g['built.chain_b'].tolist()
    assert b10 == plan['loop_b'][:len(b10)]
    eos = synthetic.PEAKED['eos']
    assert a[-1] == eos and a.count(eos) == 1 and len(a) == synthetic.PEAKED['len_a'] >= 64
    assert b15[-1] == eos and b15.count(eos) == 1 and len(b15) == synthetic.PEAKED['len_b'] >= 64
    assert eos not in b10 and len(b10) == meta['streams']['B10']['max_new_tokens']       # stopped by MaxLengthCriteria
    assert int(g['input_ids_a'][-1]) == meta['start_a'] and int(g['input_ids_b'][-1]) == plan['start_b']
    assert np.array_equal(g['input_ids_a'][:-1], g['input_ids_b'][:-1]) and g['input_ids_a'].shape == (3158,)


def test_reference_margins_are_wide():
    g, _ = load()
    for tag in ('A', 'B15', 'B10'):
        m = g[f'{tag}.margin']
        assert (m >= 1.0).mean() >= 0.9, (tag, float(m.min()), float((m >= 1.0).mean()))
        assert m.min() > 0.5, (tag, float(m.min()))
    # the penalty decided the pick at every back-edge of stream B15 (the raw arg-max was an id generated earlier)
    decided = (g['B15.raw_gap'] < 0).nonzero()[0].tolist()
    assert decided == list(synthetic.PEAKED['back_at']), decided
    assert not (g['A.raw_gap'] < 0).any() and not (g['B10.raw_gap'] < 0).any()


def test_oracle_loop_pieces_reproduce_the_reference_picks():
    """From the stored raw top-16 logits of every step: oracle.generate.apply_repetition_penalty + arg-max give the id the
    installed transformers' processor gave, and the stop rule ends each stream where the reference's criteria did."""
    from oracle.generate import apply_repetition_penalty
    g, meta = load()
    V = ModelDims.full().vocab
    for tag, cfg in meta['streams'].items():
        ids = g[f'{tag}.ids'].tolist()
        out = []
        for t in range(len(ids)):
            row = torch.full((V,), -1e30)
            row[torch.from_numpy(g[f'{tag}.top16_ids'][t])] = torch.from_numpy(g[f'{tag}.top16_logits'][t])
            row = apply_repetition_penalty(row, out, cfg['penalty'])
            nxt = int(torch.argmax(row))
            assert nxt == ids[t], (tag, t, nxt, ids[t])
            out.append(nxt)
            stop = nxt == synthetic.PEAKED['eos'] or len(out) >= cfg['max_new_tokens']
            assert stop == (t == len(ids) - 1), (tag, t)


def test_long_streams_are_the_built_walks_with_wide_margins():
    """tests/golden/peaked_long.npz (make_golden_peaked.py --long): the reference walked exactly what synthetic.PEAKED_LONG builds, for 1 024 / 85 / 397 tokens."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'peaked_long.npz'))
    meta = json.loads(bytes(g['meta']).decode())
    P = synthetic.PEAKED_LONG
    assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in P.items()} == meta['peaked_long'], \
        'synthetic.PEAKED_LONG changed: regenerate tests/golden/peaked_long.npz (scripts/make_golden_peaked.py --long)'
    plan = synthetic.peaked_plan(ModelDims.full().vocab, P['start_a'], meta['seed'], P)
    la, lb, lc = (g[f'{t}.ids'].tolist() for t in ('LA', 'LB', 'LC'))
    assert la == plan['chain_a'][:1024] and len(la) == meta['max_new_tokens'] == 1024 and P['eos'] not in la          # stopped by MaxLengthCriteria
    assert lb == plan['chain_b'] and len(lb) == 85 and len(lb) % 16 == 5 and lb[-1] == P['eos']
    assert lc == plan['chain_a'][P['long_c_entry']:] and len(lc) == 397 and lc[-1] == P['eos']
    assert g['LB.penalty_decided'].tolist() == list(P['back_at']) and g['LA.penalty_decided'].size == 0 and g['LC.penalty_decided'].size == 0
    for t in ('LA', 'LB', 'LC'):
        assert float(g[f'{t}.margin'].min()) >= 2.0, (t, float(g[f'{t}.margin'].min()))
        assert g[f'{t}.prompt'].shape[0] == P['prompt_lens'][('LA', 'LB', 'LC').index(t)]
    assert int(g['LA.prompt'][-1]) == P['start_a'] and int(g['LB.prompt'][-1]) == plan['start_b'] and int(g['LC.prompt'][-1]) == plan['chain_a'][P['long_c_entry'] - 1]
    assert os.path.getsize(os.path.join(ROOT, 'tests', 'golden', 'peaked_long.npz')) < 100 * 1024
