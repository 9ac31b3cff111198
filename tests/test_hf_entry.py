"""The reference obtains its model with `AutoModel.from_pretrained(path, torch_dtype=bf16, low_cpu_mem_usage=True,
trust_remote_code=True)` (inference.py:85-89), which imports the class the checkpoint's config.json names in `auto_map`.
With callireader_amd/hf_entry/modeling_internvl_chat.py copied over the checkpoint's file, that same call must land in
the engine's InternVLChatModel.from_pretrained.  CPU test: the engine constructor is stubbed, transformers is real."""
import json
import os
import shutil
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_automodel_auto_map_reaches_the_engine(tmp_path, monkeypatch):
    transformers = pytest.importorskip('transformers')
    ck = tmp_path / 'InternVL'
    ck.mkdir()
    shutil.copy(os.path.join(ROOT, 'callireader_amd', 'hf_entry', 'modeling_internvl_chat.py'), ck / 'modeling_internvl_chat.py')
    (ck / 'configuration_internvl_chat.py').write_text(
        'from transformers import PretrainedConfig\n\n\nclass InternVLChatConfig(PretrainedConfig):\n    model_type = "internvl_chat"\n')
    json.dump({'model_type': 'internvl_chat', 'architectures': ['InternVLChatModel'],
               'auto_map': {'AutoConfig': 'configuration_internvl_chat.InternVLChatConfig',
                            'AutoModel': 'modeling_internvl_chat.InternVLChatModel'}}, open(ck / 'config.json', 'w'))
    monkeypatch.setenv('HF_MODULES_CACHE', str(tmp_path / 'hf_modules'))
    monkeypatch.setenv('PYTHONPATH', ROOT)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import callireader_amd.modeling_internvl_chat as ours
    seen = {}

    def fake_from_pretrained(cls, path, *args, **kw):
        seen['path'], seen['kw'] = str(path), kw
        return 'engine-model'
    monkeypatch.setattr(ours.InternVLChatModel, 'from_pretrained', classmethod(fake_from_pretrained))
    model = transformers.AutoModel.from_pretrained(str(ck), torch_dtype=torch.bfloat16, low_cpu_mem_usage=True, trust_remote_code=True)
    assert model == 'engine-model'
    assert os.path.samefile(seen['path'], ck)
    assert seen['kw'].get('low_cpu_mem_usage') is True
    assert seen['kw'].get('torch_dtype', seen['kw'].get('dtype')) == torch.bfloat16
