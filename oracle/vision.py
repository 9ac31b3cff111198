"""Oracle: InternViT-300M + pixel-shuffle + mlp1 (TEST INFRASTRUCTURE ONLY).

Restates, in functional PyTorch on CPU, the eager path of
  /root/reference/InternVL/modeling_intern_vit.py
  /root/reference/InternVL/modeling_internvl_chat.py:283-319
All tensors keep the dtype of the weights (bf16 in the reference), and every
intermediate is materialised in that dtype exactly where eager PyTorch would.
`sd` is a dict keyed like the reference checkpoint.
"""
import torch
import torch.nn.functional as F


def vit_embeddings(sd, pixel_values, patch=14, image_size=448):
    """InternVisionEmbeddings.forward, modeling_intern_vit.py:167-179.

    _get_pos_embed (:159-165) interpolates the patch pos-emb bicubically in fp32
    to the (H, W) patch grid and casts back; at 448x448 the grid is the stored
    32x32 one, but the reference still runs the interpolation, so we do too.
    """
    w = sd['vision_model.embeddings.patch_embedding.weight']
    b = sd['vision_model.embeddings.patch_embedding.bias']
    cls = sd['vision_model.embeddings.class_embedding']
    pos = sd['vision_model.embeddings.position_embedding']
    dt = w.dtype
    x = F.conv2d(pixel_values, w, b, stride=patch)                  # :169
    B, _, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)                                # :171
    x = torch.cat([cls.expand(B, 1, -1).to(dt), x], dim=1)          # :172-173
    g = image_size // patch
    pe = pos[:, 1:, :].float().reshape(1, g, g, -1).permute(0, 3, 1, 2)           # :161-162
    pe = F.interpolate(pe, size=(H, W), mode='bicubic', align_corners=False)      # :163
    pe = pe.reshape(1, -1, H * W).permute(0, 2, 1).to(pos.dtype)                  # :164
    pe = torch.cat([pos[:, :1, :], pe], dim=1)                      # :174-177
    return x + pe.to(dt)                                            # :178


def vit_attention(sd, p, x, heads=16):
    """InternAttention._naive_attn, modeling_intern_vit.py:215-232 (qk_normalization False)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + 'attn.qkv.weight'], sd[p + 'attn.qkv.bias'])
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)          # :217
    q, k, v = qkv.unbind(0)
    scale = (C // heads) ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1)                        # :225  (q*scale in x.dtype)
    attn = attn.softmax(dim=-1)                                     # :226  (softmax on x.dtype tensor)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)                 # :229
    return F.linear(o, sd[p + 'attn.proj.weight'], sd[p + 'attn.proj.bias'])     # :230


def vit_mlp(sd, p, x):
    """InternMLP.forward, modeling_intern_vit.py:264-268; act = exact-erf GELU (config.json:121)."""
    h = F.linear(x, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'])
    h = F.gelu(h)
    return F.linear(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])


def vit_layer(sd, i, x, heads=16, eps=1e-6):
    """InternVisionEncoderLayer.forward, modeling_intern_vit.py:288-300 (layer_norm, drop_path 0)."""
    p = f'vision_model.encoder.layers.{i}.'
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], eps)
    x = x + vit_attention(sd, p, h, heads) * sd[p + 'ls1']          # :296
    h = F.layer_norm(x, (C,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], eps)
    x = x + vit_mlp(sd, p, h) * sd[p + 'ls2']                       # :298
    return x


def vit_forward(sd, pixel_values, n_layers, heads=16, eps=1e-6):
    """InternVisionModel.forward -> last_hidden_state, modeling_intern_vit.py:399-437."""
    if pixel_values.dim() != 4:
        raise ValueError(f'wrong pixel_values size: {pixel_values.shape}')          # :417-420
    x = vit_embeddings(sd, pixel_values)
    for i in range(n_layers):
        x = vit_layer(sd, i, x, heads, eps)
    return x


def pixel_shuffle(x, scale_factor=0.5):
    """InternVLChatModel.pixel_shuffle, ps_version 'v2', modeling_internvl_chat.py:283-297."""
    n, w, h, c = x.size()
    x = x.view(n, w, int(h * scale_factor), int(c / scale_factor))
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, int(h * scale_factor), int(w * scale_factor), int(c / (scale_factor * scale_factor)))
    x = x.permute(0, 2, 1, 3).contiguous()
    return x


def project(sd, vit_out, downsample_ratio=0.5):
    """extract_feature after the ViT, modeling_internvl_chat.py:311-318; mlp1 = :185-190."""
    x = vit_out[:, 1:, :]
    h = w = int(x.shape[1] ** 0.5)
    x = x.reshape(x.shape[0], h, w, -1)
    x = pixel_shuffle(x, downsample_ratio)
    x = x.reshape(x.shape[0], -1, x.shape[-1])
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd['mlp1.0.weight'], sd['mlp1.0.bias'], 1e-5)
    x = F.linear(x, sd['mlp1.1.weight'], sd['mlp1.1.bias'])
    x = F.gelu(x)
    return F.linear(x, sd['mlp1.3.weight'], sd['mlp1.3.bias'])


def extract_feature(sd, pixel_values, n_layers, heads=16):
    """InternVLChatModel.extract_feature (select_layer == -1), modeling_internvl_chat.py:299-319."""
    return project(sd, vit_forward(sd, pixel_values, n_layers, heads))
