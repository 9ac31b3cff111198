"""Shapes of the CalliReader image->text hot path.

Every number here comes from the reference's HF config, not from us:
  /root/reference/InternVL/config.json:11      downsample_ratio 0.5
  /root/reference/InternVL/config.json:14-103  llm_config  (InternLM2.5-7B)
  /root/reference/InternVL/config.json:114-143 vision_config (InternViT-300M)
  /root/reference/InternVL/modeling_internvl_chat.py:157  resampler depth 4
  /root/reference/models/perceiver_resampler.py:54-64     resampler dim_head 64, heads 8, 3 queries, ff x4

`ModelDims.full()` is the checkpoint's shape; `ModelDims.reduced()` keeps every
width but cuts depths so that the CPU oracle finishes in seconds (parity tests).
"""
from dataclasses import dataclass, asdict, replace

IMG_CONTEXT_TOKEN_ID = 92546   # '<IMG_CONTEXT>'  (reference added_tokens.json)
ALIGNED_TOKEN_ID = 92537       # '[UNUSED_TOKEN_140]' (modeling_internvl_chat.py:1100)
EOS_TOKEN_ID = 92542           # '<|im_end|>'     (modeling_internvl_chat.py:709,732)


@dataclass(frozen=True)
class ModelDims:
    # InternViT-300M
    image_size: int = 448
    patch_size: int = 14
    vit_hidden: int = 1024
    vit_heads: int = 16
    vit_ff: int = 4096
    vit_layers: int = 24
    vit_ln_eps: float = 1e-6
    # projector (pixel-shuffle 0.5 + mlp1)
    downsample_ratio: float = 0.5
    # PerceiverResampler ("CalliAlign")
    rs_depth: int = 4
    rs_heads: int = 8
    rs_dim_head: int = 64
    rs_queries: int = 3
    rs_ff_mult: int = 4
    # InternLM2.5-7B
    llm_hidden: int = 4096
    llm_heads: int = 32
    llm_kv_heads: int = 8
    llm_ff: int = 14336
    llm_layers: int = 32
    vocab: int = 92553
    rms_eps: float = 1e-5
    rope_theta: float = 1000000.0
    rope_factor: float = 2.0
    max_pos: int = 32768

    @property
    def n_patches(self):
        return (self.image_size // self.patch_size) ** 2          # 1024

    @property
    def vit_tokens(self):
        return self.n_patches + 1                                 # 1025

    @property
    def vit_head_dim(self):
        return self.vit_hidden // self.vit_heads                  # 64

    @property
    def tokens_per_tile(self):
        return int(self.n_patches * self.downsample_ratio ** 2)   # 256

    @property
    def proj_in(self):
        return self.vit_hidden * int(1 / self.downsample_ratio) ** 2   # 4096

    @property
    def llm_head_dim(self):
        return self.llm_hidden // self.llm_heads                  # 128

    @property
    def rs_inner(self):
        return self.rs_heads * self.rs_dim_head                   # 512

    @staticmethod
    def full():
        return ModelDims()

    @staticmethod
    def reduced(vit_layers=2, llm_layers=2, rs_depth=2, vocab=None):
        d = ModelDims()
        kw = dict(vit_layers=vit_layers, llm_layers=llm_layers, rs_depth=rs_depth)
        if vocab is not None:
            kw['vocab'] = vocab
        return replace(d, **kw)

    @staticmethod
    def from_hf_config(path, rs_depth=None):
        """Shapes from a checkpoint directory's config.json (the file AutoModel.from_pretrained reads,
        /root/reference/inference.py:85-89): depths, vocabulary, eps, RoPE base; widths must be the path's own (the HIP
        kernels are built for InternViT-300M / InternLM2.5-7B widths).  The resampler depth is not in config.json
        (modeling_internvl_chat.py:157 hard-codes 4): pass `rs_depth` or it is read off the checkpoint's key names."""
        import json
        import os
        with open(os.path.join(path, 'config.json')) as f:
            cfg = json.load(f)
        v, l = cfg['vision_config'], cfg['llm_config']
        d = ModelDims()
        want = dict(vit_hidden=v['hidden_size'], vit_heads=v['num_attention_heads'], vit_ff=v['intermediate_size'],
                    image_size=v['image_size'], patch_size=v['patch_size'], llm_hidden=l['hidden_size'],
                    llm_heads=l['num_attention_heads'], llm_kv_heads=l['num_key_value_heads'], llm_ff=l['intermediate_size'])
        for k, val in want.items():
            if getattr(d, k) != val:
                raise ValueError(f'config.json {k}={val}: this engine is built for {k}={getattr(d, k)}')
        if rs_depth is None:
            rs_depth = d.rs_depth
            idx = os.path.join(path, 'model.safetensors.index.json')
            if os.path.exists(idx):
                with open(idx) as f:
                    keys = json.load(f)['weight_map'].keys()
                layers = {int(k.split('.')[2]) for k in keys if k.startswith('resampler.layers.')}
                if layers:
                    rs_depth = max(layers) + 1
        rope = l.get('rope_scaling') or {}
        return replace(d, vit_layers=v['num_hidden_layers'], llm_layers=l['num_hidden_layers'], vocab=l['vocab_size'],
                       vit_ln_eps=v.get('layer_norm_eps', d.vit_ln_eps), rms_eps=l.get('rms_norm_eps', d.rms_eps),
                       rope_theta=float(l.get('rope_theta', d.rope_theta)), rope_factor=float(rope.get('factor', d.rope_factor)),
                       max_pos=l.get('max_position_embeddings', d.max_pos), downsample_ratio=cfg.get('downsample_ratio', d.downsample_ratio),
                       rs_depth=rs_depth)

    def asdict(self):
        return asdict(self)
