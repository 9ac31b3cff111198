"""Seeded synthetic weights in the reference checkpoint's key layout.

There are no real weights in the build or bench environment, so parity tests,
`smoke()` and `bench.py` run on random-init weights of the checkpoint's exact
architecture.  Keys are the safetensors keys of
/root/reference/InternVL/model.safetensors.index.json (vision_model.*,
mlp1.*, resampler.*, normed_emb.weight, language_model.*) plus the two side
tensors the reference loads from ./params (mu/sigma of
gauss_norm_mu_sigma.pth, modeling_internvl_chat.py:153-155) under the names
`calli.mu` / `calli.sigma`.

Every tensor is drawn from its own generator seeded by (seed, crc32(name)), so
any subset can be regenerated bit-identically on any box, on CPU.  (On the GPU
`device=` draws with the device generator: fast, deterministic per device
type, but NOT equal to the CPU draw — parity tests always draw on CPU and copy.)

Distributions are chosen so that bugs are visible (non-trivial LN gain/bias,
LayerScale ~0.1 like the real InternViT, biases non-zero); initializer ranges
follow the reference where it states them (linears N(0, 0.02):
modeling_internlm2.py:714-723; CLS/pos-emb N(0,1): modeling_intern_vit.py:146-157;
resampler queries N(0,1): perceiver_resampler.py:66).
"""
import zlib
import torch

from .config import ModelDims


def _gen(name, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def _randn(name, shape, std, seed, device, dtype, mean=0.0):
    g = _gen(name, seed, device)
    t = torch.empty(shape, device=device, dtype=torch.float32)
    t.normal_(mean, std, generator=g)
    return t.to(dtype)


def vit_keys(d: ModelDims):
    P = d.patch_size
    out = {
        'vision_model.embeddings.class_embedding': ((1, 1, d.vit_hidden), 1.0, 0.0),
        'vision_model.embeddings.position_embedding': ((1, d.vit_tokens, d.vit_hidden), 1.0, 0.0),
        'vision_model.embeddings.patch_embedding.weight': ((d.vit_hidden, 3, P, P), 0.02, 0.0),
        'vision_model.embeddings.patch_embedding.bias': ((d.vit_hidden,), 0.02, 0.0),
    }
    for i in range(d.vit_layers):
        p = f'vision_model.encoder.layers.{i}.'
        out[p + 'norm1.weight'] = ((d.vit_hidden,), 0.1, 1.0)
        out[p + 'norm1.bias'] = ((d.vit_hidden,), 0.05, 0.0)
        out[p + 'norm2.weight'] = ((d.vit_hidden,), 0.1, 1.0)
        out[p + 'norm2.bias'] = ((d.vit_hidden,), 0.05, 0.0)
        out[p + 'attn.qkv.weight'] = ((3 * d.vit_hidden, d.vit_hidden), 0.02, 0.0)
        out[p + 'attn.qkv.bias'] = ((3 * d.vit_hidden,), 0.02, 0.0)
        out[p + 'attn.proj.weight'] = ((d.vit_hidden, d.vit_hidden), 0.02, 0.0)
        out[p + 'attn.proj.bias'] = ((d.vit_hidden,), 0.02, 0.0)
        out[p + 'mlp.fc1.weight'] = ((d.vit_ff, d.vit_hidden), 0.02, 0.0)
        out[p + 'mlp.fc1.bias'] = ((d.vit_ff,), 0.02, 0.0)
        out[p + 'mlp.fc2.weight'] = ((d.vit_hidden, d.vit_ff), 0.02, 0.0)
        out[p + 'mlp.fc2.bias'] = ((d.vit_hidden,), 0.02, 0.0)
        out[p + 'ls1'] = ((d.vit_hidden,), 0.05, 0.15)
        out[p + 'ls2'] = ((d.vit_hidden,), 0.05, 0.15)
    return out


def mlp1_keys(d: ModelDims):
    return {
        'mlp1.0.weight': ((d.proj_in,), 0.1, 1.0),
        'mlp1.0.bias': ((d.proj_in,), 0.05, 0.0),
        'mlp1.1.weight': ((d.llm_hidden, d.proj_in), 0.02, 0.0),
        'mlp1.1.bias': ((d.llm_hidden,), 0.02, 0.0),
        'mlp1.3.weight': ((d.llm_hidden, d.llm_hidden), 0.02, 0.0),
        'mlp1.3.bias': ((d.llm_hidden,), 0.02, 0.0),
    }


def resampler_keys(d: ModelDims):
    D = d.llm_hidden
    out = {'resampler.learns': ((d.rs_queries, D), 1.0, 0.0),
           'resampler.norm.weight': ((D,), 0.1, 1.0),
           'resampler.norm.bias': ((D,), 0.05, 0.0)}
    for i in range(d.rs_depth):
        a = f'resampler.layers.{i}.0.'
        f = f'resampler.layers.{i}.1.net.'
        out[a + 'norm_media.weight'] = ((D,), 0.1, 1.0)
        out[a + 'norm_media.bias'] = ((D,), 0.05, 0.0)
        out[a + 'norm_learns.weight'] = ((D,), 0.1, 1.0)
        out[a + 'norm_learns.bias'] = ((D,), 0.05, 0.0)
        out[a + 'to_q.weight'] = ((d.rs_inner, D), 0.02, 0.0)
        out[a + 'to_kv.weight'] = ((2 * d.rs_inner, D), 0.02, 0.0)
        out[a + 'to_out.weight'] = ((D, d.rs_inner), 0.02, 0.0)
        out[f + '0.weight'] = ((D,), 0.1, 1.0)
        out[f + '0.bias'] = ((D,), 0.05, 0.0)
        out[f + '1.weight'] = ((D * d.rs_ff_mult, D), 0.02, 0.0)
        out[f + '1.bias'] = ((D * d.rs_ff_mult,), 0.02, 0.0)
        out[f + '3.weight'] = ((D, D * d.rs_ff_mult), 0.02, 0.0)
        out[f + '3.bias'] = ((D,), 0.02, 0.0)
    return out


def vq_keys(d: ModelDims):
    return {
        'normed_emb.weight': ((d.vocab, d.llm_hidden), 1.0, 0.0),
        'calli.mu': ((d.vocab, 1), 0.002, 0.0),
        'calli.sigma': ((d.vocab, 1), 0.002, 0.02),
    }


def llm_keys(d: ModelDims):
    D = d.llm_hidden
    hd = d.llm_head_dim
    out = {
        'language_model.model.tok_embeddings.weight': ((d.vocab, D), 0.02, 0.0),
        'language_model.model.norm.weight': ((D,), 0.1, 1.0),
        'language_model.output.weight': ((d.vocab, D), 0.02, 0.0),
    }
    for i in range(d.llm_layers):
        p = f'language_model.model.layers.{i}.'
        out[p + 'attention_norm.weight'] = ((D,), 0.1, 1.0)
        out[p + 'ffn_norm.weight'] = ((D,), 0.1, 1.0)
        out[p + 'attention.wqkv.weight'] = (((d.llm_heads + 2 * d.llm_kv_heads) * hd, D), 0.02, 0.0)
        out[p + 'attention.wo.weight'] = ((D, D), 0.02, 0.0)
        out[p + 'feed_forward.w1.weight'] = ((d.llm_ff, D), 0.02, 0.0)
        out[p + 'feed_forward.w3.weight'] = ((d.llm_ff, D), 0.02, 0.0)
        out[p + 'feed_forward.w2.weight'] = ((D, d.llm_ff), 0.02, 0.0)
    return out


PARTS = {
    'vit': vit_keys,
    'mlp1': mlp1_keys,
    'resampler': resampler_keys,
    'vq': vq_keys,
    'llm': llm_keys,
}


def key_specs(dims: ModelDims, parts=('vit', 'mlp1', 'resampler', 'vq', 'llm')):
    specs = {}
    for p in parts:
        specs.update(PARTS[p](dims))
    return specs


def make_tensor(name, spec, seed=0, device='cpu', dtype=torch.bfloat16):
    shape, std, mean = spec
    return _randn(name, shape, std, seed, device, dtype, mean)


def make_state_dict(dims: ModelDims, parts=('vit', 'mlp1', 'resampler', 'vq', 'llm'),
                    seed=0, device='cpu', dtype=torch.bfloat16, outlier_shift=0):
    """Full dict.  For the 7.7 B-parameter LLM prefer `iter_state_dict` (streams)."""
    return {k: outlier_transform(k, make_tensor(k, s, seed, device, dtype), dims, outlier_shift) for k, s in key_specs(dims, parts).items()}


def iter_state_dict(dims: ModelDims, parts=('vit', 'mlp1', 'resampler', 'vq', 'llm'),
                    seed=0, device='cpu', dtype=torch.bfloat16, outlier_shift=0):
    for k, s in key_specs(dims, parts).items():
        yield k, outlier_transform(k, make_tensor(k, s, seed, device, dtype), dims, outlier_shift)


# ---- "outlier" checkpoints: the SAME function with the channel / row statistics real checkpoints have ---------------------------------------------
# Random-Gaussian weights have no outlier channels, so they cannot tell a good fp8 scaling scheme from a bad one (round-4 verdict, item 4a).  Real LLM / ViT
# checkpoints do: a few normalised-activation channels carry values tens to hundreds of times the rest (large norm gains against small consuming weight
# columns), and some weight rows are far larger than others.  This variant re-scales the seed-0 tensors by POWERS OF TWO in matched pairs:
#   * norm gain (and LayerNorm bias) of a few channels x 2^shift, the consuming linear's columns of those channels x 2^-shift:
#       LLM attention_norm -> wqkv, ffn_norm -> w1 / w3, final norm -> output; ViT norm1 -> qkv, norm2 -> fc1; mlp1.0 -> mlp1.1;
#   * every OUTLIER['w3_rows_every']-th row of w3 (the linear branch of SwiGLU) x 2^shift, the matching w2 columns x 2^-shift: outlier channels in w2's INPUT.
# A power-of-two factor commutes with every bf16 / fp32 rounding (no overflow or underflow at these magnitudes), so the network's function is unchanged BIT FOR
# BIT in the reference's bf16 arithmetic -- tests/golden/full_depth.npz, generated by the reference on the plain checkpoint, is also the golden of every
# outlier variant (tests/test_oracle_golden.py checks the oracle on both, tests/test_gpu_fp8_mfma.py the HIP bf16 path) -- while every row-wise maximum an
# fp8 quantiser takes is now dominated by the outlier channels (x 32 at shift 5, x 1024 at shift 10).
OUTLIER = dict(llm_channels=(5, 700, 1403, 2222, 3071, 4000), vit_channels=(3, 257, 640, 1001), w3_rows_every=1021)


def outlier_transform(name, t, dims: ModelDims, shift, cfg=OUTLIER):
    """`t` (a tensor of the plain checkpoint under key `name`) -> the outlier variant's tensor; shift = 0 returns `t` itself."""
    if not shift:
        return t
    up, dn = 2.0 ** shift, 2.0 ** -shift
    lc = torch.tensor(cfg['llm_channels'], device=t.device)
    vc = torch.tensor(cfg['vit_channels'], device=t.device)
    t = t.clone()
    if name.startswith('language_model.'):
        if name.endswith('attention_norm.weight') or name.endswith('ffn_norm.weight') or name == 'language_model.model.norm.weight':
            t[lc] *= up
        elif name.endswith('attention.wqkv.weight') or name.endswith('feed_forward.w1.weight') or name == 'language_model.output.weight':
            t[:, lc] *= dn
        elif name.endswith('feed_forward.w3.weight'):
            t[:, lc] *= dn
            t[::cfg['w3_rows_every']] *= up
        elif name.endswith('feed_forward.w2.weight'):
            t[:, ::cfg['w3_rows_every']] *= dn
    elif name.startswith('vision_model.encoder.layers.'):
        if name.endswith(('norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias')):
            t[vc] *= up
        elif name.endswith('attn.qkv.weight') or name.endswith('mlp.fc1.weight'):
            t[:, vc] *= dn
    elif name in ('mlp1.0.weight', 'mlp1.0.bias'):
        t[lc] *= up
    elif name == 'mlp1.1.weight':
        t[:, lc] *= dn
    return t


def make_pixels(n_tiles, seed=0, device='cpu', dtype=torch.bfloat16, size=448):
    """ImageNet-normalised pixels are ~N(0,1)."""
    return _randn('pixel_values', (n_tiles, 3, size, size), 1.0, seed, device, dtype)


# ---- OrderFormer (params/orderformer.pth, models/model.py:206-233,530-552) --------------------------------------
ORDERFORMER = dict(max_nums=50, input_dim=4, model_dim=256, num_heads=8, num_layers=4, ff=2048)


def orderformer_keys(cfg=ORDERFORMER):
    """state_dict keys of the reference's `Transformer` (nn.Linear embedding, nn.TransformerEncoder of post-norm
    nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward=2048, relu), nn.Linear decoder, no final norm).
    The template layer `encoder_layer.*` is a registered sub-module too, so a strict load needs it although only
    its deep copies `transformer_encoder.layers.N.*` run."""
    d, ff = cfg['model_dim'], cfg['ff']
    specs = {'embedding.weight': ((d, cfg['input_dim']), 0.7, 0.0), 'embedding.bias': ((d,), 0.1, 0.0),
             'decoder.weight': ((1, d), 0.08, 0.0), 'decoder.bias': ((1,), 0.1, 0.0)}
    prefixes = ['encoder_layer.'] + [f'transformer_encoder.layers.{i}.' for i in range(cfg['num_layers'])]
    for p in prefixes:
        specs[p + 'self_attn.in_proj_weight'] = ((3 * d, d), d ** -0.5, 0.0)
        specs[p + 'self_attn.in_proj_bias'] = ((3 * d,), 0.05, 0.0)
        specs[p + 'self_attn.out_proj.weight'] = ((d, d), d ** -0.5, 0.0)
        specs[p + 'self_attn.out_proj.bias'] = ((d,), 0.05, 0.0)
        specs[p + 'linear1.weight'] = ((ff, d), d ** -0.5, 0.0)
        specs[p + 'linear1.bias'] = ((ff,), 0.05, 0.0)
        specs[p + 'linear2.weight'] = ((d, ff), ff ** -0.5, 0.0)
        specs[p + 'linear2.bias'] = ((d,), 0.05, 0.0)
        for n in ('norm1', 'norm2'):
            specs[p + n + '.weight'] = ((d,), 0.1, 1.0)
            specs[p + n + '.bias'] = ((d,), 0.05, 0.0)
    return specs


def make_orderformer_state_dict(seed=0, device='cpu', dtype=torch.bfloat16, cfg=ORDERFORMER):
    return {k: _randn('orderformer.' + k, shape, std, seed, device, dtype, mean=mean)
            for k, (shape, std, mean) in orderformer_keys(cfg).items()}


# ---- margin-controlled ("peaked") language model ---------------------------------------------------------------
# Random-init weights make the logits nearly flat (typical top-2 margin 0.1-0.9 at |logit| <= 5.5): the unfavourable
# case for a token-exactness test, because any second bf16 implementation flips such near-ties.  This checkpoint keeps the
# architecture, every key and every tensor of the seed-0 set EXCEPT three families, so that the reference's own greedy loop
# (InternVL/modeling_internvl_chat.py:1111-1120 -> modeling_internlm2.py:1022-1149) decodes with a top-2 margin of ~2-4 at |logit| ~ 12:
#   * wo / w2 of every layer: the seed-0 tensors times 2^-PEAKED['branch_shift'] (exact in bf16) -- the 64 residual
#     branches together add about as much to the residual stream as the token embedding it started from, instead of
#     swamping it 100:1;
#   * output.weight = N(0, noise_std) + a * E[T1^-1] + runner * a * E[T2^-1] + a few override rows, E = tok_embeddings:
#     after token v the row of T1(v) scores ~ a * |e_v|^2 / rms(h), the row of T2(v) `runner` times that;
#   * T1 is one cycle through the whole vocabulary (a permutation) laid out so that the walk from `start_a` reaches EOS as
#     its len_a-th token; the walk from `start_b` carries BACK-EDGES (override rows scoring `back` x the successor's): at those
#     steps the un-penalised arg-max is a token generated a few steps earlier, so with repetition_penalty 1.0 the stream
#     loops until max_new_tokens, and with 1.5 (score / 1.5 for generated ids: transformers 4.45.2
#     RepetitionPenaltyLogitsProcessor) it walks on and stops at an EOS override as its len_b-th token.
PEAKED = dict(branch_shift=9, a=0.25, runner=0.7, back=1.2, noise_std=0.002, len_a=72, len_b=80,
              back_at=(9, 17, 26, 34, 45, 58, 66), back_hop=4, runner_shift=40503, eos=92542, start_b=2000)


# The LONG streams (round-5 verdict, item 5): the same construction at reduced depth (2 layers, full width, full vocabulary) with a walk A that does not reach EOS
# within the API's default max_new_tokens = 1024 (inference.py:92-96) and a walk B whose EOS override comes as its 85th token (85 = 5 mod 16: the engine looks for
# EOS every 16 steps).  A third stream enters walk A `long_c_entry` tokens in and reaches A's EOS as its (len_a - long_c_entry)-th token.
PEAKED_LONG = dict(PEAKED, len_a=1400, len_b=85, back_at=(9, 17, 26, 34, 45, 58, 66, 77), long_c_entry=1003, llm_layers=2, start_a=3000,
                   prompt_lens=(333, 77, 200))


def peaked_plan(vocab, start_a, seed=0, cfg=PEAKED):
    """The walk tables of the peaked checkpoint.  Returns dict(t1, t2 (LongTensor[vocab]), overrides [(v, u, strength)],
    chain_a, chain_b (ids each walk is BUILT to generate, EOS included), loop_b (what chain B's prompt generates without a
    penalty: the first back-edge closes a cycle))."""
    eos, start_b = cfg['eos'], cfg['start_b']
    assert len({start_a, start_b, eos}) == 3
    g = torch.Generator()
    g.manual_seed(int(seed) * 7919 + 20251002)
    order = [t for t in torch.randperm(vocab, generator=g).tolist() if t not in (start_a, start_b, eos)]
    la, lb = cfg['len_a'], cfg['len_b']
    seq = [start_a] + order[:la - 1] + [eos, start_b] + order[la - 1:]
    assert len(seq) == vocab
    seq_t = torch.tensor(seq, dtype=torch.long)
    t1 = torch.empty(vocab, dtype=torch.long)
    t1[seq_t] = torch.roll(seq_t, -1)
    t2 = torch.empty(vocab, dtype=torch.long)
    t2[seq_t] = torch.roll(seq_t, -cfg['runner_shift'])
    pos_b = la + 1                                       # index of start_b in seq
    z = seq[pos_b:pos_b + lb + 1]                        # z[0] = start_b, z[k] = k-th token of walk B
    a = cfg['a']
    overrides = [(z[k], z[k - cfg['back_hop']], cfg['back'] * a) for k in cfg['back_at']]
    overrides.append((z[lb - 1], eos, cfg['back'] * a))
    chain_a = seq[1:la + 1]
    chain_b = z[1:lb] + [eos]
    k0, hop = cfg['back_at'][0], cfg['back_hop']
    cyc = z[k0 - hop:k0 + 1]
    loop_b = z[1:k0 + 1] + [cyc[i % len(cyc)] for i in range(64)]
    assert chain_a[-1] == eos and len(chain_a) == la and len(chain_b) == lb
    return dict(t1=t1, t2=t2, overrides=overrides, chain_a=chain_a, chain_b=chain_b, loop_b=loop_b, start_b=start_b)


def peaked_output_weight(dims: ModelDims, start_a, seed=0, cfg=PEAKED, dtype=torch.bfloat16):
    """language_model.output.weight of the peaked checkpoint (CPU; ~5 GB of fp32 temporaries)."""
    plan = peaked_plan(dims.vocab, start_a, seed, cfg)
    name = 'language_model.output.weight'
    E = make_tensor('language_model.model.tok_embeddings.weight', llm_keys(dims)['language_model.model.tok_embeddings.weight'],
                    seed, 'cpu', torch.bfloat16).float()
    W = _randn(name + '#peaked', (dims.vocab, dims.llm_hidden), cfg['noise_std'], seed, 'cpu', torch.float32)
    W.index_add_(0, plan['t1'], E * cfg['a'])
    W.index_add_(0, plan['t2'], E * (cfg['a'] * cfg['runner']))
    for v, u, s in plan['overrides']:
        W[u] += E[v] * s
    return W.to(dtype)


def iter_peaked_llm(dims: ModelDims, start_a, seed=0, cfg=PEAKED, device='cpu', dtype=torch.bfloat16):
    """(key, tensor) over the language-model keys of the peaked checkpoint, in `llm_keys` order."""
    scale = 2.0 ** -cfg['branch_shift']
    for k, s in llm_keys(dims).items():
        if k == 'language_model.output.weight':
            yield k, peaked_output_weight(dims, start_a, seed, cfg, dtype).to(device)
        elif k.endswith('attention.wo.weight') or k.endswith('feed_forward.w2.weight'):
            yield k, make_tensor(k, s, seed, device, dtype) * scale
        else:
            yield k, make_tensor(k, s, seed, device, dtype)


def peaked_prompt_embeds(n_vit_rows, n_ref_rows, hidden=4096, seed=0, dtype=torch.bfloat16):
    """Stand-ins for extract_feature's rows and calli_align's pseudo tokens in the peaked run's prompt (CPU draw)."""
    return (_randn('peaked.vit_embeds', (n_vit_rows, hidden), 0.05, seed, 'cpu', dtype),
            _randn('peaked.reference_embeds', (n_ref_rows, hidden), 0.02, seed, 'cpu', dtype))
