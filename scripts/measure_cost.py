#!/usr/bin/env python3
"""parallel.measure_cost on this GPU at full InternVL2-8B shapes (random-init weights), next to the table it replaces; --fp8: with the fp8 options on."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from callireader_amd.config import ModelDims  # noqa: E402
from callireader_amd.modeling_internvl_chat import InternVLChatModel  # noqa: E402
from callireader_amd import parallel  # noqa: E402

fp8 = '--fp8' in sys.argv
m = InternVLChatModel.from_synthetic(ModelDims.full(), seed=0, max_tokens=4096, max_pages=64)
if fp8:
    m.engine.enable_fp8_mfma(True, level=2)
    m.engine.enable_fp8_decode(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
cost = parallel.measure_cost(m)
dt = time.perf_counter() - t0
t0 = time.perf_counter()
cost2 = parallel.measure_cost(m)
dt2 = time.perf_counter() - t0
table = parallel.default_cost(m)
out = {'fp8': fp8, 'first_call_s': round(dt, 2), 'second_call_s': round(dt2, 2), 'measured': cost, 'measured_again': cost2,
       'table': {k: table[k] for k in ('tile_ms', 'char_tile_ms', 'chunk_ms', 'prefill_ms_per_token', 'decode_ms', 'decode_ctx_tokens')}}
for name, c in (('measured', cost), ('table', table)):
    pl = parallel.plan_balanced(64, 8, 11, 96, 3164, 128, cost=c)
    out[f'plan_64_pages_8_ranks_{name}'] = {'k': pl['k'], 'char_counts': pl['char_counts'], 'predicted_step_ms': pl['predicted_step_ms'], 'predicted_even_ms': pl['predicted_even_ms']}
print(json.dumps(out, indent=1))
