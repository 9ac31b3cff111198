"""Checkpoint formats of the reference -> cr_load_weight calls.

  * HF sharded safetensors: <path>/model.safetensors.index.json + shards (629 tensors: vision_model.*,
    language_model.*, mlp1.*, resampler.*, normed_emb.weight) — what AutoModel.from_pretrained reads in
    /root/reference/inference.py:85-89;
  * ./params/gauss_norm_mu_sigma.pth  {'weight': (vocab,2)} -> mu = [:,0], sigma = [:,1]
    (/root/reference/InternVL/modeling_internvl_chat.py:153-155);
  * optional ./params/callialign.pth (resampler, DDP 'module.' prefix stripped, 'model_state_dict' unwrapped:
    /root/reference/models/model.py:100-118) and ./params/gauss_norm.pth (normalised table,
    modeling_internvl_chat.py:195-197) when those tensors are not in the safetensors shards.
Tensors stream shard by shard; each is uploaded and released, so host memory stays at one tensor.
"""
import json
import os

import torch


def strip_ddp(state_dict):
    if 'model_state_dict' in state_dict:
        state_dict = state_dict['model_state_dict']
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}


def iter_safetensors(path):
    from safetensors import safe_open
    index = os.path.join(path, 'model.safetensors.index.json')
    if os.path.exists(index):
        with open(index) as f:
            weight_map = json.load(f)['weight_map']
        shards = sorted(set(weight_map.values()))
    else:
        shards = sorted(f for f in os.listdir(path) if f.endswith('.safetensors'))
    if not shards:
        raise FileNotFoundError(f'no safetensors shards under {path}')
    for shard in shards:
        fp = os.path.join(path, shard)
        if not os.path.exists(fp):
            raise FileNotFoundError(fp)
        with safe_open(fp, framework='pt', device='cpu') as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def load_checkpoint(engine, path, params_dir='./params'):
    seen = set()
    for k, t in iter_safetensors(path):
        engine.load_weight(k, t.to(torch.bfloat16) if t.is_floating_point() else t)
        seen.add(k)
    ms = os.path.join(params_dir, 'gauss_norm_mu_sigma.pth')
    if not os.path.exists(ms):
        raise FileNotFoundError(f'{ms} (mu/sigma of the normalised token table) is required')
    w = torch.load(ms, map_location='cpu')['weight']
    engine.load_weight('calli.mu', w[:, 0].reshape(-1, 1).contiguous())
    engine.load_weight('calli.sigma', w[:, 1].reshape(-1, 1).contiguous())
    if not any(k.startswith('resampler.') for k in seen):
        sd = strip_ddp(torch.load(os.path.join(params_dir, 'callialign.pth'), map_location='cpu', weights_only=False))
        for k, t in sd.items():
            engine.load_weight('resampler.' + k, t.to(torch.bfloat16))
    if 'normed_emb.weight' not in seen:
        sd = torch.load(os.path.join(params_dir, 'gauss_norm.pth'), map_location='cpu', weights_only=True)
        engine.load_weight('normed_emb.weight', sd['weight'].to(torch.bfloat16))
    return seen
