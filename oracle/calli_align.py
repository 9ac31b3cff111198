"""Oracle: PerceiverResampler ("CalliAlign"), cosine VQ, de-normalisation (TEST INFRASTRUCTURE ONLY).

Restates
  /root/reference/models/perceiver_resampler.py:8-100,130-141
  /root/reference/models/similarity.py:9-27
  /root/reference/InternVL/modeling_internvl_chat.py:602-640 (calli_align tail)
"""
import torch
import torch.nn.functional as F


def perceiver_attention(sd, p, x, learns, heads=8, dim_head=64):
    """PerceiverAttention.forward, perceiver_resampler.py:28-51."""
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd[p + 'norm_media.weight'], sd[p + 'norm_media.bias'])
    learns = F.layer_norm(learns, (D,), sd[p + 'norm_learns.weight'], sd[p + 'norm_learns.bias'])
    b = x.shape[0]
    q = F.linear(learns, sd[p + 'to_q.weight'])                                # :35
    kv_input = torch.cat((x, learns), dim=-2)                                  # :38
    k, v = F.linear(kv_input, sd[p + 'to_kv.weight']).chunk(2, dim=-1)        # :39

    def split(t):                                                              # 'b n (h d) -> b h n d'
        return t.reshape(b, t.shape[1], heads, dim_head).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    q = q * (dim_head ** -0.5)                                                 # :43
    sim = torch.einsum('bhid,bhjd->bhij', q, k)                                # :46
    sim = sim - sim.amax(dim=-1, keepdim=True)                                 # :47
    attn = sim.softmax(dim=-1)                                                 # :48
    out = torch.einsum('bhij,bhjd->bhid', attn, v)                             # :50
    out = out.permute(0, 2, 1, 3).reshape(b, out.shape[2], heads * dim_head)   # :51
    return F.linear(out, sd[p + 'to_out.weight'])


def feed_forward(sd, p, x):
    """FeedForward.forward, perceiver_resampler.py:130-141 (LN -> Linear -> GELU -> Linear)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + '0.weight'], sd[p + '0.bias'])
    h = F.linear(h, sd[p + '1.weight'], sd[p + '1.bias'])
    h = F.gelu(h)
    return F.linear(h, sd[p + '3.weight'], sd[p + '3.bias'])


def resampler_forward(sd, x, depth, heads=8, dim_head=64):
    """PerceiverResampler.forward, perceiver_resampler.py:81-100."""
    b = x.shape[0]
    learns = sd['resampler.learns'].unsqueeze(0).expand(b, -1, -1)             # :92
    for i in range(depth):
        learns = perceiver_attention(sd, f'resampler.layers.{i}.0.', x, learns, heads, dim_head) + learns   # :97
        learns = feed_forward(sd, f'resampler.layers.{i}.1.net.', learns) + learns                          # :98
    D = x.shape[-1]
    return F.layer_norm(learns, (D,), sd['resampler.norm.weight'], sd['resampler.norm.bias'])               # :100


def vq_cos_sim(table, x, use_dynamic_p=False):
    """vq_cos_sim, similarity.py:9-27.  table = normed_emb.weight (V, D); x (B, n, D)."""
    input_norm = F.normalize(x, p=2, dim=2)
    embedding_norm = F.normalize(table, p=2, dim=1)
    similarity = torch.matmul(input_norm, embedding_norm.t())
    cos_sim_values, indices = similarity.max(dim=2)
    if use_dynamic_p:
        return indices.squeeze(), cos_sim_values.squeeze()
    return indices.squeeze()


def denormalise(x, indices, table, mu, sigma, drop_zero=False, hard_vq=False, cos=None, thresh=0.5):
    """calli_align tail, modeling_internvl_chat.py:602-640.

    x (B,3,D) resampler output; indices (B,3); mu/sigma (V,1).
    Returns (back_to_origin_flat (n,D), indices).
    """
    if hard_vq:
        below = (cos <= thresh).to(torch.bfloat16).unsqueeze(-1)               # :612
        x = x * (1 - below) + table[indices] * below                            # :614
    flat = x.reshape(-1, x.shape[-1])
    fidx = indices.reshape(-1)
    if drop_zero:                                                               # :620-630
        keep = fidx != 0
        flat = flat[keep]
        fidx = fidx[keep]
    s = sigma[fidx].expand(-1, flat.shape[-1])
    m = mu[fidx].expand(-1, flat.shape[-1])
    return flat * s + m, indices


def bf16_step(v):
    """One bf16 step (2^-7 of the binade) at the magnitude of `v`; 0 for v == 0."""
    import math
    v = abs(float(v))
    return 2.0 ** (math.floor(math.log2(v)) - 7) if v > 0 else 0.0


def vq_tie_rule(ref_sim_ref_id, ref_sim_other_id, other_sim_ref_id, other_sim_other_id):
    """THE rule for a cosine-VQ index that differs from the reference's (one function for tests/test_gpu_calli.py, tests/test_gpu_full_depth.py and
    scripts/real_checkpoint_parity.py; round-4 verdict, item 7).  similarity.py:17-21 takes the arg-max of a bf16 similarity matrix, so index work is
    exact work and another implementation's index may differ only at a MEASURED tie:
      (a) the reference's own similarities (the bf16 matrix of similarity.py:19, as floats) at its pick and at the other pick are at most ONE bf16
          step apart (and ordered: its pick is its maximum), and
      (b) the other implementation's similarities at the same two rows straddle that gap: its pick scores >= the reference's pick there.
    `ref_sim_other_id` = None means the other pick is not even among the reference's recorded candidates: no tie.  Returns (ok, gap, step)."""
    step = bf16_step(ref_sim_ref_id)
    if ref_sim_other_id is None:
        return False, float('inf'), step
    gap = float(ref_sim_ref_id) - float(ref_sim_other_id)
    ok = 0.0 <= gap <= step and float(other_sim_other_id) >= float(other_sim_ref_id)
    return ok, gap, step
