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
                    seed=0, device='cpu', dtype=torch.bfloat16):
    """Full dict.  For the 7.7 B-parameter LLM prefer `iter_state_dict` (streams)."""
    return {k: make_tensor(k, s, seed, device, dtype) for k, s in key_specs(dims, parts).items()}


def iter_state_dict(dims: ModelDims, parts=('vit', 'mlp1', 'resampler', 'vq', 'llm'),
                    seed=0, device='cpu', dtype=torch.bfloat16):
    for k, s in key_specs(dims, parts).items():
        yield k, make_tensor(k, s, seed, device, dtype)


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
