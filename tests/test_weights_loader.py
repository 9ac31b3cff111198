"""Checkpoint formats: HF sharded safetensors + params/*.pth -> the engine's load_weight calls.
CPU-only: a recording fake engine stands in for the GPU context."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from callireader_amd import weights


class Recorder:
    def __init__(self):
        self.got = {}

    def load_weight(self, name, t):
        self.got[name] = t.clone()


def make_ckpt(tmp, with_resampler=True, with_table=True):
    a = {'vision_model.embeddings.class_embedding': torch.randn(1, 1, 8).bfloat16(), 'mlp1.0.weight': torch.randn(8).float()}
    b = {'language_model.output.weight': torch.randn(4, 8).bfloat16()}
    if with_resampler:
        b['resampler.learns'] = torch.randn(3, 8).bfloat16()
    if with_table:
        b['normed_emb.weight'] = torch.randn(5, 8).bfloat16()
    save_file(a, os.path.join(tmp, 'model-00001-of-00002.safetensors'))
    save_file(b, os.path.join(tmp, 'model-00002-of-00002.safetensors'))
    wm = {k: 'model-00001-of-00002.safetensors' for k in a}
    wm.update({k: 'model-00002-of-00002.safetensors' for k in b})
    json.dump({'metadata': {}, 'weight_map': wm}, open(os.path.join(tmp, 'model.safetensors.index.json'), 'w'))
    params = os.path.join(tmp, 'params')
    os.makedirs(params)
    torch.save({'weight': torch.arange(10, dtype=torch.float32).reshape(5, 2)}, os.path.join(params, 'gauss_norm_mu_sigma.pth'))
    return a, b, params


def test_sharded_safetensors_and_mu_sigma(tmp_path):
    a, b, params = make_ckpt(str(tmp_path))
    rec = Recorder()
    seen = weights.load_checkpoint(rec, str(tmp_path), params)
    assert seen == set(a) | set(b)
    assert rec.got['mlp1.0.weight'].dtype == torch.bfloat16            # fp32 tensors are cast like torch_dtype=bf16 does
    assert torch.equal(rec.got['language_model.output.weight'], b['language_model.output.weight'])
    assert rec.got['calli.mu'].shape == (5, 1) and rec.got['calli.mu'].flatten().tolist() == [0, 2, 4, 6, 8]
    assert rec.got['calli.sigma'].flatten().tolist() == [1, 3, 5, 7, 9] and rec.got['calli.sigma'].dtype == torch.float32


def test_side_files_fill_missing_tensors(tmp_path):
    a, b, params = make_ckpt(str(tmp_path), with_resampler=False, with_table=False)
    # DDP-prefixed, wrapped resampler checkpoint as models/model.py:100-118 handles it
    torch.save({'model_state_dict': {'module.learns': torch.ones(3, 8), 'module.norm.weight': torch.ones(8)}},
               os.path.join(params, 'callialign.pth'))
    torch.save({'weight': torch.full((5, 8), 2.0)}, os.path.join(params, 'gauss_norm.pth'))
    rec = Recorder()
    weights.load_checkpoint(rec, str(tmp_path), params)
    assert 'resampler.learns' in rec.got and 'resampler.norm.weight' in rec.got
    assert rec.got['normed_emb.weight'].dtype == torch.bfloat16 and float(rec.got['normed_emb.weight'][0, 0]) == 2.0


def test_missing_files_raise(tmp_path):
    with pytest.raises(FileNotFoundError):
        weights.load_checkpoint(Recorder(), str(tmp_path), str(tmp_path))
    a, b, params = make_ckpt(str(tmp_path))
    os.remove(os.path.join(params, 'gauss_norm_mu_sigma.pth'))
    with pytest.raises(FileNotFoundError):
        weights.load_checkpoint(Recorder(), str(tmp_path), params)


def test_strip_ddp():
    sd = weights.strip_ddp({'model_state_dict': {'module.a': 1, 'b': 2}})
    assert sd == {'a': 1, 'b': 2}
