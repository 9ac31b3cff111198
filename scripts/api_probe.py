#!/usr/bin/env python3
"""The api_level block alone (benchlib/api.py) on a fresh synthetic model at full dims: development probe for what separates the API path from the synthetic step.
env CR_PIPE_CHECK_EVERY=n: the decode thread's EOS look every n steps."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd.modeling_internvl_chat import InternVLChatModel
from benchlib.api import api_level
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = InternVLChatModel.from_synthetic(ModelDims.full(), seed=0, max_tokens=3420, max_pages=64)
ms = torch.cuda.Stream(priority=0)
torch.cuda.set_stream(ms)
r = api_level(m, ROOT, pages=64, batches=int(os.environ.get('BATCHES', '5')), new_tokens=128, folder_pages=64)
r.pop('what', None); r.pop('host_note', None); r.pop('idle_note', None); r.get('folder_rec', {}).pop('what', None)
print(json.dumps(r))
