#!/usr/bin/env python3
"""Development aid: does the poisoned build (-DCR_POISON: every LDS piece is filled with bf16 NaNs right before its LDS-DMA refill) SEE a
schedule that is actually broken (-DCR_BREAK_WAIT: no wait after the cold-start fills, no counted wait in the main loop)?  Runs large integer GEMMs whose operands miss L2, with a copy hog on a second stream raising the memory
latency, and counts outputs that differ from the exact integer answer.  usage: CR_HIP_LIB=ab/lib<variant>.so hazard_teeth.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from callireader_amd import engine as E  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
out = {'lib': os.environ.get('CR_HIP_LIB', 'default'), 'cases': []}
hogA = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
hogB = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
for (M, N, K, kern) in [(8192, 8192, 8192, 5), (8192, 8192, 8192, 6), (32768, 4096, 1024, 5), (16384, 4096, 14336, 6)]:
    A = torch.randint(-1, 2, (M, K), generator=g).to(torch.bfloat16).to(dev)
    W = torch.randint(-1, 2, (N, K), generator=g).to(torch.bfloat16).to(dev)
    ref = (A.float() @ W.float().t()).to(torch.bfloat16)
    bad = nan = 0
    for it in range(int(os.environ.get('CR_TEETH_ITERS', '4'))):
        with torch.cuda.stream(side):
            for _ in range(40):
                hogB.copy_(hogA)
        o = E.op_gemm(0, A, W, kernel=kern)
        torch.cuda.synchronize()
        nan += int(torch.isnan(o.float()).sum())
        bad += int((o != ref).sum())
    out['cases'].append({'M': M, 'N': N, 'K': K, 'pin': kern, 'wrong': bad, 'nan': nan})
print(json.dumps(out))
