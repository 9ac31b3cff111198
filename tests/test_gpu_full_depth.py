"""FULL-DEPTH parity on a real MI355X against vectors the REFERENCE itself produced.

tests/golden/full_depth.npz was written in the build container by scripts/make_golden_full_depth.py, which imports the
reference's own classes from /root/reference: InternVisionModel (24 layers) + InternVLChatModel.extract_feature on 2
tiles, and InternLM2ForCausalLM (32 layers, 92 553-row vocabulary, eager attention) on a 300-token prompt followed by
8 greedy steps of a hand loop over `forward`.  Weights are the seeded `callireader_amd.synthetic` tensors, regenerated
here bit for bit on the CPU and uploaded.  The file also holds the reference's OWN bf16-vs-fp32 difference at these
depths (same modules cast to fp32, same bf16 weight values), which is what the tolerances below are stated against:
a second bf16 implementation with another accumulation order cannot sit closer to the bf16 reference than the bf16
reference sits to exact arithmetic.

Token bar: the HIP path's free-running greedy ids must equal the reference's.  A difference is only accepted at a step
where the reference's own top-2 margin is below the measured logit difference bound (a tie the reference itself would
resolve differently under any re-association), and then the test prints the step and that margin and walks on
teacher-forced.  Everything measured is written to profiles/round4/full_depth_parity.json (round 3's run: profiles/round3/) (gpurun_out/ on the GPU box
as well) so the numbers quoted in DESIGN.md have an artifact.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from callireader_amd.config import ModelDims
from callireader_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'full_depth.npz')
RESULTS = {}


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


def subsample(t, step, n):
    return t.detach().float().reshape(-1)[::step][:n]


def bits_to_f32(u16):
    return torch.from_numpy(u16.astype(np.int32) << 16).view(torch.float32)


# Token bar.  A greedy pick that differs from the reference's is accepted ONLY as a measured near-tie:
#   * the HIP pick must be one of the reference's own top-16 candidates of that step (its reference logit is then known);
#   * gap = ref[ref_id] - ref[hip_id] (>= the reference's top-2 margin) must be covered by the logit differences MEASURED AT THOSE
#     TWO IDS, gap <= |d(ref_id)| + |d(hip_id)| -- i.e. the two HIP logits straddle, nothing is excused by a row-wide bound;
#   * each of the two differences must itself be within the reference's own bf16-vs-fp32 max |d| (the yardstick), and
#   * gap <= TIE_CAP: every mismatch ever recorded on these weights sits at a reference margin <= 0.156 and every step with a
#     margin >= 0.19 matches (profiles/round2/full_depth_parity.json), so 0.25 = 4 bf16 steps at |logit| 4..8 is the cap
#     (round 2 excused up to 2 x the row's max |d| ~ 0.7).
# Free-running token-exactness itself is demonstrated where the reference's margins allow it: tests/test_gpu_peaked.py.
TIE_CAP = 0.25


def check_near_tie(where, got, ref_id, hip_id, ref_logit_of, noise_abs):
    assert hip_id in ref_logit_of, f'{where}: HIP picked {hip_id}, which is not among the reference\'s top-16 candidates'
    gap = ref_logit_of[ref_id] - ref_logit_of[hip_id]
    d_ref, d_hip = float(got[ref_id]) - ref_logit_of[ref_id], float(got[hip_id]) - ref_logit_of[hip_id]
    rec = {'gap': gap, 'd_ref_id': d_ref, 'd_hip_id': d_hip}
    print(f'  {where}: pick differs: reference gap {gap:.4f}, measured d(ref id) {d_ref:+.4f}, d(hip id) {d_hip:+.4f}')
    assert gap <= d_hip - d_ref + 1e-6, (where, rec)            # a literal straddle (oracle/generate.py: near_tie_straddles), signs and all
    assert max(abs(d_ref), abs(d_hip)) <= noise_abs, (where, rec)
    assert gap <= TIE_CAP, (where, rec)
    return rec


def _dump():
    for d in (os.path.join(ROOT, 'profiles', 'round5'), os.path.join(ROOT, 'gpurun_out')):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, 'full_depth_parity.json'), 'w') as f:
                json.dump(RESULTS, f, indent=1)
        except OSError:
            pass


@pytest.fixture(scope='module')
def gold():
    g = np.load(GOLD)
    meta = json.loads(bytes(g['meta']).decode())
    assert meta['seed'] == 0 and meta['tiles'] == 2
    return g, meta


@pytest.fixture(scope='module')
def engine():
    from callireader_amd.engine import Engine
    dims = ModelDims.full()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    eng = Engine(dims, max_pos=4096)
    t0 = time.time()
    for parts in (('vit', 'mlp1'), ('resampler', 'vq'), ('llm',)):
        for k, v in synthetic.iter_state_dict(dims, parts=parts, seed=0):     # CPU draw == the golden script's weights
            eng.load_weight(k, v)
    eng.load_rope()
    eng.finalize()
    RESULTS['weights_s'] = round(time.time() - t0, 1)
    yield eng
    eng.close()


def test_vit_24_layers_and_projector_vs_reference(gold, engine):
    g, meta = gold
    px = synthetic.make_pixels(2, seed=meta['pixels_seed'])
    last = engine.vit_forward(px.cuda())
    feat = engine.extract_feature(px.cuda())
    torch.cuda.synchronize()
    noise = float(g['vit24.ref_bf16_vs_fp32_rel_l2'])           # the reference against itself in fp32
    out = {}
    for name, t in (('last', last), ('feat', feat)):
        ref = torch.from_numpy(g[f'vit24.{name}.sample'])
        got = subsample(t.cpu(), int(g[f'vit24.{name}.step']), ref.numel())
        assert int(g[f'vit24.{name}.numel']) == t.numel()
        out[name] = {'rel_l2': rel_l2(got, ref), 'max_abs': float((got - ref).abs().max()), 'ref_max_abs': float(ref.abs().max()),
                     'sum_rel': abs(float(t.double().sum().item()) - float(g[f'vit24.{name}.sum'])) / float(g[f'vit24.{name}.abssum'])}
    rows = torch.from_numpy(g['vit24.feat_rows'])
    out['feat_rows_rel_l2'] = rel_l2(feat[:, :4, :].float().cpu(), rows)
    out['reference_bf16_vs_fp32_rel_l2'] = noise
    RESULTS['vit24'] = out
    _dump()
    print('full-depth vision:', json.dumps(out))
    assert torch.isfinite(feat.float()).all()
    # bound = the reference's own bf16 noise at this depth (1.25e-2 measured by the golden script); the HIP path
    # measured 5e-3..7e-3
    assert out['feat']['rel_l2'] <= noise, out
    assert out['last']['rel_l2'] <= noise, out
    assert out['feat_rows_rel_l2'] <= noise, out
    assert out['feat']['sum_rel'] <= 2e-3 and out['last']['sum_rel'] <= 2e-3, out


def test_llm_32_layers_full_vocab_vs_reference(gold, engine):
    g, meta = gold
    S, steps = meta['prompt_tokens'], meta['steps']
    gen = torch.Generator().manual_seed(meta['prompt_seed'])
    emb = (torch.randn(1, S, 4096, generator=gen) * 0.02).to(torch.bfloat16)
    ref_logits = bits_to_f32(g['llm32.logits_bf16_bits'])       # [steps + 1][vocab]
    ref_ids = g['llm32.greedy_tokens'].tolist()
    margins = g['llm32.top2_margin'].tolist()
    noise_l2, noise_abs = float(g['llm32.ref_bf16_vs_fp32_rel_l2']), float(g['llm32.ref_bf16_vs_fp32_max_abs'])
    assert ref_logits.shape == (steps + 1, ModelDims.full().vocab)

    # (1) free-running greedy: prefill + `steps` decode steps, nothing forced
    kv = engine.kv_alloc(1, 512)
    engine.prefill(kv, 0, emb.cuda())
    for _ in range(steps):
        engine.decode(kv, [0])
    free_ids = kv.generated(0)[:steps + 1]
    first_div = next((i for i, (a, b) in enumerate(zip(free_ids, ref_ids)) if a != b), None)

    # (2) teacher-forced along the reference's ids: every step's logits against the reference's
    kv.reset(0)
    lg = engine.prefill(kv, 0, emb.cuda(), want_logits=True)
    per_step, picks, rows_kept = [], [], []
    for t in range(steps + 1):
        got, ref = lg.float().cpu().reshape(-1), ref_logits[t]
        rows_kept.append(got)
        picked = kv.generated(0)[t]
        picks.append(picked)
        per_step.append({'rel_l2': rel_l2(got, ref), 'max_abs': float((got - ref).abs().max()),
                         'ref_margin': margins[t], 'ref_id': ref_ids[t], 'hip_id': picked,
                         'hip_gap_at_ref_ids': float(got[picked] - got[ref_ids[t]])})
        if t < steps:
            lg = engine.decode(kv, [0], force_tokens=torch.tensor([ref_ids[t]]), want_logits=True)
    torch.cuda.synchronize()
    # KV rows the reference kept
    out = {'prompt_tokens': S, 'steps': steps, 'reference_ids': ref_ids, 'hip_free_running_ids': free_ids,
           'first_divergence': first_div, 'per_step': per_step,
           'reference_bf16_vs_fp32': {'rel_l2': noise_l2, 'max_abs': noise_abs},
           'worst_rel_l2': max(p['rel_l2'] for p in per_step), 'worst_max_abs': max(p['max_abs'] for p in per_step)}
    RESULTS['llm32'] = out
    _dump()
    print('full-depth LLM:', json.dumps({k: v for k, v in out.items() if k != 'per_step'}))
    for t, p in enumerate(per_step):
        print(f'  step {t}: rel-L2 {p["rel_l2"]:.3e} max|d| {p["max_abs"]:.3f} ref id {p["ref_id"]} (margin {p["ref_margin"]:.4f}) hip id {p["hip_id"]}')
    # logits: within the reference's own bf16-vs-fp32 noise at this depth
    assert out['worst_rel_l2'] <= noise_l2, out['worst_rel_l2']
    assert out['worst_max_abs'] <= max(noise_abs, 0.12), out['worst_max_abs']
    # tokens: exact, or a reference near-tie (margin below the logit difference actually measured at that step)
    top_ids, top_val = g['llm32.top16_ids'], g['llm32.top16_logits']
    for t, p in enumerate(per_step):
        if p['hip_id'] != p['ref_id']:
            p['near_tie'] = check_near_tie(f'llm32 step {t}', rows_kept[t], p['ref_id'], p['hip_id'],
                                           dict(zip(top_ids[t].tolist(), top_val[t].tolist())), noise_abs)
    _dump()
    if first_div is None:
        assert free_ids == ref_ids
    else:
        assert per_step[first_div]['hip_id'] != per_step[first_div]['ref_id'], 'free-running and teacher-forced picks disagree'
    kv.free()


def test_llm_kv_rows_vs_reference(gold, engine):
    """K of layer 0 at the last prompt position (after RoPE at position 299) and V of layer 31 at position 0, as the
    reference's tuple cache holds them (modeling_internlm2.py:383-388); and prefill is deterministic."""
    g, meta = gold
    S = meta['prompt_tokens']
    gen = torch.Generator().manual_seed(meta['prompt_seed'])
    emb = (torch.randn(1, S, 4096, generator=gen) * 0.02).to(torch.bfloat16)
    kv = engine.kv_alloc(2, 512)
    a = engine.prefill(kv, 0, emb.cuda(), want_logits=True)
    b = engine.prefill(kv, 1, emb.cuda(), want_logits=True)
    k0 = kv.read(0, 1, S - 1, 0).float().cpu()
    v31 = kv.read(31, 1, 0, 1).float().cpu()
    torch.cuda.synchronize()
    assert torch.equal(a, b), 'prefill is not deterministic'
    rk, rv = torch.from_numpy(g['llm32.k0_last']), torch.from_numpy(g['llm32.v31_first'])
    out = {'k_layer0_last_rel_l2': rel_l2(k0, rk), 'v_layer31_first_rel_l2': rel_l2(v31, rv)}
    RESULTS['llm32_kv'] = out
    _dump()
    print('full-depth KV rows:', out)
    assert out['k_layer0_last_rel_l2'] <= 1e-2, out          # layer 0: one RMSNorm + GEMM + RoPE deep
    assert out['v_layer31_first_rel_l2'] <= float(g['llm32.ref_bf16_vs_fp32_rel_l2']) * 2, out   # 31 layers of residual stream behind it
    kv.free()


def test_llm_layerwise_error_budget(gold, engine):
    """Where the logit difference comes from: the residual stream before layer 0 and after each of the 32 layers (three prompt
    rows), HIP against the reference's own hidden states (tests/golden/full_depth_layers.npz, scripts/make_golden_layers.py: forward
    hooks on the reference's decoder layers), next to the reference's OWN bf16-vs-fp32 distance at the same layer.  Written to
    profiles/round4/full_depth_parity.json (round 3's run: profiles/round3/) (key llm32_layers); bound: at every layer the HIP path sits no further from the bf16
    reference than the bf16 reference sits from its fp32 self."""
    g, meta = gold
    gl = np.load(os.path.join(ROOT, 'tests', 'golden', 'full_depth_layers.npz'))
    rows = gl['layers.rows'].tolist()
    ref = bits_to_f32(gl['layers.bf16_bits'])                   # [33][3][4096]
    yard = gl['layers.ref_bf16_vs_fp32_rel_l2'].tolist()
    S = meta['prompt_tokens']
    gen = torch.Generator().manual_seed(meta['prompt_seed'])
    emb = (torch.randn(1, S, 4096, generator=gen) * 0.02).to(torch.bfloat16)
    probe = engine.hidden_probe(rows=S)
    kv = engine.kv_alloc(1, 512)
    try:
        engine.prefill(kv, 0, emb.cuda())
        torch.cuda.synchronize()
        got = probe[:, rows, :].float().cpu()
    finally:
        engine.hidden_probe(None)
        kv.free()
    per = [{'after_layer': l, 'hip_vs_reference_rel_l2': rel_l2(got[l], ref[l]), 'reference_bf16_vs_fp32_rel_l2': yard[l],
            'residual_rms': float(gl['layers.ref_rms'][l])} for l in range(ref.shape[0])]
    RESULTS['llm32_layers'] = {'rows': rows, 'per_layer': per,
                               'worst_ratio_to_yardstick': max(p['hip_vs_reference_rel_l2'] / max(p['reference_bf16_vs_fp32_rel_l2'], 1e-12) for p in per[1:])}
    _dump()
    for p in per:
        print(f"  residual after layer {p['after_layer']:2d}: HIP vs reference {p['hip_vs_reference_rel_l2']:.3e}   reference bf16 vs fp32 {p['reference_bf16_vs_fp32_rel_l2']:.3e}")
    assert per[0]['hip_vs_reference_rel_l2'] == 0.0              # the stack's input is the prompt itself
    for p in per[1:]:
        assert p['hip_vs_reference_rel_l2'] <= p['reference_bf16_vs_fp32_rel_l2'], p


def test_llm_extra_prompts_token_agreement(gold, engine):
    """Two more prompts (64 and 513 tokens, 12 greedy steps each), teacher-forced along the reference's ids: per step the
    reference's top-16 logits and a stride-8 sample of the row are compared, and the greedy pick is counted."""
    g, meta = gold
    noise_l2, noise_abs = float(g['llm32.ref_bf16_vs_fp32_rel_l2']), float(g['llm32.ref_bf16_vs_fp32_max_abs'])
    summary = []
    for i, (seed, tokens, steps) in enumerate(meta['extra_prompts']):
        tag = f'llm32.extra{i}'
        gen = torch.Generator().manual_seed(seed)
        emb = (torch.randn(1, tokens, 4096, generator=gen) * 0.02).to(torch.bfloat16)
        ref_ids, margins = g[f'{tag}.greedy_tokens'].tolist(), g[f'{tag}.top2_margin'].tolist()
        top_ids, top_val = torch.from_numpy(g[f'{tag}.top16_ids']), torch.from_numpy(g[f'{tag}.top16_logits'])
        strided = bits_to_f32(g[f'{tag}.logits_stride8_bf16_bits'])
        kv = engine.kv_alloc(1, 1024)
        lg = engine.prefill(kv, 0, emb.cuda(), want_logits=True)
        rows, kept = [], []
        for t in range(steps + 1):
            got = lg.float().cpu().reshape(-1)
            kept.append(got)
            picked = kv.generated(0)[t]
            rows.append({'rel_l2_stride8': rel_l2(got[::8], strided[t]), 'max_abs_top16': float((got[top_ids[t]] - top_val[t]).abs().max()),
                         'max_abs_stride8': float((got[::8] - strided[t]).abs().max()),
                         'ref_id': ref_ids[t], 'hip_id': picked, 'ref_margin': margins[t]})
            if t < steps:
                lg = engine.decode(kv, [0], force_tokens=torch.tensor([ref_ids[t]]), want_logits=True)
        torch.cuda.synchronize()
        kv.free()
        agree = sum(r['ref_id'] == r['hip_id'] for r in rows)
        summary.append({'prompt_tokens': tokens, 'steps': steps, 'picks_equal': agree, 'picks': len(rows),
                        'worst_rel_l2': max(r['rel_l2_stride8'] for r in rows), 'worst_max_abs': max(r['max_abs_stride8'] for r in rows),
                        'differing': [{'step': t, **r} for t, r in enumerate(rows) if r['ref_id'] != r['hip_id']]})
        for t, r in enumerate(rows):
            assert r['rel_l2_stride8'] <= noise_l2, (tag, t, r)
            if r['ref_id'] != r['hip_id']:
                r['near_tie'] = check_near_tie(f'{tag} step {t}', kept[t], r['ref_id'], r['hip_id'],
                                               dict(zip(top_ids[t].tolist(), top_val[t].tolist())), noise_abs)
        summary[-1]['differing'] = [{'step': t, **r} for t, r in enumerate(rows) if r['ref_id'] != r['hip_id']]
    RESULTS['llm32_extra'] = summary
    _dump()
    print('full-depth LLM, extra prompts:', json.dumps(summary))


def test_config1_example_page_end_to_end_vs_reference(gold, engine):
    """BASELINE config 1 at full depth on the reference's own example page (tests/golden/config1_full_depth.npz, written by
    scripts/make_golden_config1.py through the reference's modules): 11 page tiles + 96 character tiles -> ViT 24L + mlp1 ->
    PerceiverResampler -> cosine VQ -> denormalised pseudo tokens -> the 3 158-id prompt of the reference's tokenizer with
    both masked overwrites -> InternLM2 32L prefill + 15 greedy steps, teacher-forced along the reference's ids."""
    from PIL import Image
    from callireader_amd import preprocess
    g1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'config1_full_depth.npz'))
    g, _ = gold
    noise_vit, noise_llm = float(g['vit24.ref_bf16_vs_fp32_rel_l2']), float(g['llm32.ref_bf16_vs_fp32_rel_l2'])
    img = Image.open(os.path.join(ROOT, 'tests', 'golden', 'example0.jpg')).convert('RGB')
    boxes = preprocess.boxes_from_labelme(json.load(open(os.path.join(ROOT, 'tests', 'golden', 'example0_boxes.json'))))
    page_px = preprocess.load_image(img).to(torch.bfloat16)
    arr = np.array(img)
    char_px = torch.cat([preprocess.load_image_2(Image.fromarray(arr[y1:y2, x1:x2])).to(torch.bfloat16) for x1, y1, x2, y2 in boxes])
    assert page_px.shape[0] == 11 and char_px.shape[0] == 96

    feat_page = engine.extract_feature(page_px.cuda())
    feat_chars = engine.extract_feature(char_px.cuda())
    rs = engine.resample(feat_chars)
    idx = engine.vq(rs)
    back = engine.denorm(rs, idx)
    torch.cuda.synchronize()
    out = {}
    for name, t in (('feat_page', feat_page), ('feat_chars', feat_chars), ('resampler', rs), ('pseudo', back)):
        ref = torch.from_numpy(g1[f'{name}.sample'])
        assert int(g1[f'{name}.numel']) == t.numel(), name
        got = subsample(t.cpu(), int(g1[f'{name}.step']), ref.numel())
        out[name] = {'rel_l2': rel_l2(got, ref), 'max_abs': float((got - ref).abs().max()), 'ref_max_abs': float(ref.abs().max())}
    ref_idx = torch.from_numpy(g1['vq.indices'])
    gap = torch.from_numpy(g1['vq.top2_cos'])
    gap = (gap[..., 0] - gap[..., 1]).reshape(-1)
    neq = (idx.cpu().reshape(-1) != ref_idx.reshape(-1))
    out['vq'] = {'rows': int(neq.numel()), 'equal': int((~neq).sum()), 'reference_top2_gap_min': float(gap.min()),
                 'largest_reference_gap_where_different': float(gap[neq].max()) if neq.any() else 0.0, 'differing': []}
    if neq.any():
        # the ONE rule for a differing VQ index (oracle/calli_align.py: vq_tie_rule, shared with tests/test_gpu_calli.py and scripts/real_checkpoint_parity.py):
        # the reference's similarities at the two rows (its recorded top-8 candidates, vq.top8_*) at most one bf16 step apart AND the HIP tiled GEMM's own
        # similarities, on the HIP path's resampler row and the normalised table, straddling that gap
        from oracle import calli_align
        from callireader_amd import engine as E, synthetic
        from test_gpu_calli import hip_similarities
        top_ids, top_cos = torch.from_numpy(g1['vq.top8_ids']).reshape(-1, 8), torch.from_numpy(g1['vq.top8_cos']).reshape(-1, 8)
        table = synthetic.make_state_dict(ModelDims.full(), parts=('vq',), seed=0)['normed_emb.weight']
        tn = torch.nn.functional.normalize(table, p=2, dim=1)
        xn = torch.nn.functional.normalize(rs.cpu().reshape(-1, rs.shape[-1]), p=2, dim=1)
        for r in neq.nonzero().reshape(-1).tolist():
            i_o, i_h = int(ref_idx.reshape(-1)[r]), int(idx.cpu().reshape(-1)[r])
            cand = dict(zip(top_ids[r].tolist(), top_cos[r].tolist()))
            s_hip = hip_similarities(E, xn[r], tn, i_o, i_h)
            ok, vgap, step = calli_align.vq_tie_rule(cand[i_o], cand.get(i_h), s_hip[0], s_hip[1])
            out['vq']['differing'].append({'row': r, 'ref_id': i_o, 'hip_id': i_h, 'reference_gap': vgap, 'one_bf16_step': step, 'hip_similarities': s_hip, 'measured_tie': ok})

    ids = torch.from_numpy(g1['input_ids'])
    assert int((ids == 92546).sum()) == 11 * 256 and int((ids == 92537).sum()) == back.numel() // 4096
    emb = engine.embed_splice(ids, vit_embeds=feat_page, ref_embeds=back)
    ref_ids, margins = g1['greedy_tokens'].tolist(), g1['top2_margin'].tolist()
    top_ids, top_val = torch.from_numpy(g1['top16_ids']), torch.from_numpy(g1['top16_logits'])
    strided = bits_to_f32(g1['logits_stride8_bf16_bits'])
    steps = len(ref_ids) - 1
    kv = engine.kv_alloc(1, 3328)
    lg = engine.prefill(kv, 0, emb, want_logits=True)
    rows, kept = [], []
    for t in range(steps + 1):
        got = lg.float().cpu().reshape(-1)
        kept.append(got)
        picked = kv.generated(0)[t]
        rows.append({'rel_l2_stride8': rel_l2(got[::8], strided[t]), 'max_abs_stride8': float((got[::8] - strided[t]).abs().max()),
                     'max_abs_top16': float((got[top_ids[t]] - top_val[t]).abs().max()), 'ref_id': ref_ids[t], 'hip_id': picked,
                     'ref_margin': margins[t]})
        if t < steps:
            lg = engine.decode(kv, [0], force_tokens=torch.tensor([ref_ids[t]]), want_logits=True)
    torch.cuda.synchronize()
    kv.free()
    out['llm'] = {'prompt_tokens': int(ids.numel()), 'steps': steps, 'picks_equal': sum(r['ref_id'] == r['hip_id'] for r in rows), 'picks': len(rows),
                  'worst_rel_l2': max(r['rel_l2_stride8'] for r in rows), 'worst_max_abs': max(r['max_abs_stride8'] for r in rows),
                  'differing': [{'step': t, **r} for t, r in enumerate(rows) if r['ref_id'] != r['hip_id']]}
    RESULTS['config1_example_page'] = out
    _dump()
    print('config 1, full depth:', json.dumps(out))
    for name in ('feat_page', 'feat_chars'):
        assert out[name]['rel_l2'] <= noise_vit, (name, out[name])
    # the resampler adds 4 bf16 layers on top of the visual features; the pseudo tokens are its output times sigma plus mu
    assert out['resampler']['rel_l2'] <= 2 * noise_vit and out['pseudo']['rel_l2'] <= 2 * noise_vit, out
    # VQ: equal, or a measured tie by the one rule (vq_tie_rule) -- round 4 accepted any differing index whose reference top-2 gap was <= 0.002
    assert all(d['measured_tie'] for d in out['vq']['differing']), out['vq']
    for t, r in enumerate(rows):
        assert r['rel_l2_stride8'] <= noise_llm, (t, r)
        if r['ref_id'] != r['hip_id']:
            r['near_tie'] = check_near_tie(f'config 1 step {t}', kept[t], r['ref_id'], r['hip_id'],
                                           dict(zip(top_ids[t].tolist(), top_val[t].tolist())), float(g['llm32.ref_bf16_vs_fp32_max_abs']))


def test_fp8_mfma_option_at_full_depth_is_recorded(gold, engine):
    """The fp8 matrix-core option (OFF by default) at full depth against the same reference vectors, both levels: recorded next to
    the bf16 numbers in profiles/round4/full_depth_parity.json (round 3's run: profiles/round3/).  It is a THROUGHPUT option: e4m3 keeps 3 mantissa bits, every fp8 linear
    adds ~5 % of independent relative noise and on random-init weights 24 / 32 layers of it accumulate, so the numbers below are far
    outside the bf16 path's (and the greedy pick usually differs); what it costs on a real checkpoint is what
    `evaluate.py --compare_fp8` measures.  The assertions are sanity bounds only, and level 1 (norm-fed linears only) must sit closer
    to the reference than level 2."""
    g, meta = gold
    px = synthetic.make_pixels(2, seed=meta['pixels_seed'])
    S = meta['prompt_tokens']
    gen = torch.Generator().manual_seed(meta['prompt_seed'])
    emb = (torch.randn(1, S, 4096, generator=gen) * 0.02).to(torch.bfloat16)
    ref_logits = bits_to_f32(g['llm32.logits_bf16_bits'])[0]
    ref = torch.from_numpy(g['vit24.feat.sample'])
    out = {}
    for level in (1, 2):
        engine.enable_fp8_mfma(True, level=level)
        try:
            feat = engine.extract_feature(px.cuda())
            kv = engine.kv_alloc(1, 512)
            lg = engine.prefill(kv, 0, emb.cuda(), want_logits=True).float().cpu().reshape(-1)
            kv.free()
        finally:
            engine.enable_fp8_mfma(False)
        torch.cuda.synchronize()
        got = subsample(feat.cpu(), int(g['vit24.feat.step']), ref.numel())
        assert torch.isfinite(feat.float()).all() and torch.isfinite(lg).all()
        out[f'level{level}'] = {'extract_feature_rel_l2_vs_reference': rel_l2(got, ref), 'prefill_logits_rel_l2_vs_reference': rel_l2(lg, ref_logits),
                                'prefill_pick_equal': bool(int(lg.argmax()) == int(ref_logits.argmax()))}
    out['bf16_path_for_comparison'] = {'extract_feature': RESULTS.get('vit24', {}).get('feat', {}).get('rel_l2'),
                                       'logits_worst': RESULTS.get('llm32', {}).get('worst_rel_l2')}
    out['reading'] = 'not parity-preserving on these weights at either level; a throughput option whose gate is evaluate.py --compare_fp8 on a real checkpoint'
    RESULTS['fp8_mfma_full_depth'] = out
    _dump()
    print('fp8 MFMA option at full depth:', json.dumps(out))
    for level in (1, 2):
        o = out[f'level{level}']
        assert o['extract_feature_rel_l2_vs_reference'] <= 0.2 and o['prefill_logits_rel_l2_vs_reference'] <= 0.8, out
    assert out['level1']['prefill_logits_rel_l2_vs_reference'] <= out['level2']['prefill_logits_rel_l2_vs_reference'], out


def test_vit_config2_properties(engine):
    """BASELINE config 2 at full size (32 tiles, 24 layers): finite, deterministic, and invariant to how the tiles are
    chunked -- 32 at once == 31 + 1 == 16 + 16, bit for bit (a tile's result never depends on its batch)."""
    px = synthetic.make_pixels(32, seed=0).cuda()
    full = engine.vit_forward(px)
    again = engine.vit_forward(px)
    torch.cuda.synchronize()
    assert torch.isfinite(full.float()).all()
    assert torch.equal(full, again)
    a = torch.cat([engine.vit_forward(px[:31]), engine.vit_forward(px[31:])])
    b = torch.cat([engine.vit_forward(px[:16]), engine.vit_forward(px[16:])])
    torch.cuda.synchronize()
    assert torch.equal(full, a) and torch.equal(full, b)
    RESULTS['vit_config2_properties'] = {'tiles': 32, 'finite': True, 'deterministic': True, 'chunking_invariant': True}
    _dump()
