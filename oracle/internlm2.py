"""Oracle: InternLM2 decoder, eager attention, tuple KV cache (TEST INFRASTRUCTURE ONLY).

Restates /root/reference/InternVL/modeling_internlm2.py (eager path, batch 1..B,
no padding inside a row: the hot path always passes an all-ones attention_mask).
"""
import math
import torch
import torch.nn.functional as F


def rms_norm(x, weight, eps=1e-5):
    """InternLM2RMSNorm.forward, modeling_internlm2.py:138-143 (cast to input dtype BEFORE * weight)."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)


def rope_tables(head_dim=128, max_pos=32768, base=1000000.0, seq_len=None, factor=2.0):
    """InternLM2DynamicNTKScalingRotaryEmbedding._set_cos_sin_cache, modeling_internlm2.py:213-229.

    Built at init for seq_len = max_position_embeddings in the default dtype
    (fp32); the NTK base rescale only applies when seq_len > max_pos (:216-221).
    Returns fp32 (cos, sin) of shape (seq_len, head_dim); callers cast to the
    activation dtype as rotary_emb.forward does (:177-180).
    """
    seq_len = max_pos if seq_len is None else seq_len
    if seq_len > max_pos:
        base = base * ((factor * seq_len / max_pos) - (factor - 1)) ** (head_dim / (head_dim - 2))
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))     # :154 / :220
    t = torch.arange(seq_len).to(inv_freq.dtype)
    freqs = torch.einsum('i,j->ij', t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    """modeling_internlm2.py:233-237."""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """apply_rotary_pos_emb, modeling_internlm2.py:241-247 (unsqueeze_dim=1)."""
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def causal_mask(q_len, past_len, dtype):
    """_make_causal_mask + all-ones _expand_mask, modeling_internlm2.py:96-125,830-851."""
    if q_len <= 1:
        return torch.zeros(1, 1, q_len, q_len + past_len, dtype=dtype)
    m = torch.full((q_len, q_len), torch.finfo(dtype).min)
    cond = torch.arange(q_len)
    m.masked_fill_(cond < (cond + 1).view(q_len, 1), 0)
    m = m.to(dtype)
    if past_len > 0:
        m = torch.cat([torch.zeros(q_len, past_len, dtype=dtype), m], dim=-1)
    return m[None, None]


def attention(sd, p, x, position_ids, past_kv, cos_t, sin_t, n_heads=32, n_kv=8):
    """InternLM2Attention.forward, modeling_internlm2.py:341-426."""
    B, q_len, D = x.shape
    hd = D // n_heads
    groups = n_heads // n_kv
    qkv = F.linear(x, sd[p + 'attention.wqkv.weight'])                                    # :359
    qkv = qkv.view(B, q_len, n_kv, groups + 2, hd)                                        # :361-366
    q = qkv[..., :groups, :].reshape(B, q_len, n_heads, hd).transpose(1, 2)               # :368-369,373
    k = qkv[..., -2, :].transpose(1, 2)                                                   # :370,374
    v = qkv[..., -1, :].transpose(1, 2)                                                   # :371,375
    kv_len = q_len + (past_kv[0].shape[-2] if past_kv is not None else 0)
    cos = cos_t[:kv_len].to(x.dtype)                                                      # :177-180
    sin = sin_t[:kv_len].to(x.dtype)
    q, k = apply_rope(q, k, cos, sin, position_ids)                                       # :381
    if past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)                                             # :385
        v = torch.cat([past_kv[1], v], dim=2)
    present = (k, v)
    kr = k[:, :, None].expand(B, n_kv, groups, kv_len, hd).reshape(B, n_heads, kv_len, hd)   # repeat_kv :268-277
    vr = v[:, :, None].expand(B, n_kv, groups, kv_len, hd).reshape(B, n_heads, kv_len, hd)
    w = torch.matmul(q, kr.transpose(2, 3)) / math.sqrt(hd)                               # :393
    w = w + causal_mask(q_len, kv_len - q_len, x.dtype)                                   # :406
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)                             # :409
    o = torch.matmul(w, vr)                                                               # :410
    o = o.transpose(1, 2).contiguous().reshape(B, q_len, D)                               # :418-419
    return F.linear(o, sd[p + 'attention.wo.weight']), present                            # :421


def mlp(sd, p, x):
    """InternLM2MLP.forward, modeling_internlm2.py:261-264."""
    return F.linear(F.silu(F.linear(x, sd[p + 'feed_forward.w1.weight'])) *
                    F.linear(x, sd[p + 'feed_forward.w3.weight']),
                    sd[p + 'feed_forward.w2.weight'])


def decoder_layer(sd, i, x, position_ids, past_kv, cos_t, sin_t, n_heads=32, n_kv=8, eps=1e-5):
    """InternLM2DecoderLayer.forward, modeling_internlm2.py:621-681."""
    p = f'language_model.model.layers.{i}.'
    h, present = attention(sd, p, rms_norm(x, sd[p + 'attention_norm.weight'], eps),
                           position_ids, past_kv, cos_t, sin_t, n_heads, n_kv)
    x = x + h
    x = x + mlp(sd, p, rms_norm(x, sd[p + 'ffn_norm.weight'], eps))
    return x, present


def model_forward(sd, n_layers, inputs_embeds=None, input_ids=None, past=None, rope=None,
                  n_heads=32, n_kv=8, eps=1e-5, all_logits=True):
    """InternLM2ForCausalLM.forward, modeling_internlm2.py:1022-1110 + InternLM2Model.forward :854-984.

    Returns (logits fp32 (B, S or 1, V), new_past).  `all_logits=False` computes
    only the last row (the only row greedy decoding reads); the reference always
    computes all rows (:1081), row values are identical.
    """
    if inputs_embeds is None:
        inputs_embeds = F.embedding(input_ids, sd['language_model.model.tok_embeddings.weight'])   # :901
    B, S, D = inputs_embeds.shape
    past_len = past[0][0].shape[2] if past is not None else 0                                        # :889-891
    position_ids = torch.arange(past_len, past_len + S).unsqueeze(0)                                 # :893-898
    if rope is None:
        rope = rope_tables(D // n_heads)
    x = inputs_embeds
    new_past = []
    for i in range(n_layers):
        x, present = decoder_layer(sd, i, x, position_ids, past[i] if past is not None else None,
                                   rope[0], rope[1], n_heads, n_kv, eps)
        new_past.append(present)
    x = rms_norm(x, sd['language_model.model.norm.weight'], eps)                                     # :970
    if not all_logits:
        x = x[:, -1:, :]
    logits = F.linear(x, sd['language_model.output.weight']).float()                                 # :1081-1082
    return logits, tuple(new_past)
