#!/usr/bin/env python3
"""bench.py's `roofline.by_kernel` block alone (benchlib/kernels.py): every kernel of the path at its launch shape of the 64-page step against its own roofline."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchlib.kernels import by_kernel
r = by_kernel(torch.device('cuda', 0))
for k, v in r.items():
    print(f"{k:20s} {v['ms']:9.4f} ms  {v['achieved']:8.1f} {v['unit']:8s} frac {v['frac']:.4f}")
print(json.dumps(r))
