#!/usr/bin/env python3
"""Golden vectors for the calli_align TAIL (hard VQ, drop_zero, sigma/mu de-normalisation) produced by the reference's
own statements (build container only).

The tail is inline in `InternVLChatModel.calli_align` (InternVL/modeling_internvl_chat.py:602-640) and the method cannot
run as a whole here (YOLO, files, CUDA).  This script takes the method's source at generation time, keeps the top-level
statements that follow `outs = vq_cos_sim(...)` up to the `return`, compiles them into a function of
(self, outs, output, use_hard_vector_quant, drop_zero) and runs THAT on seeded inputs.  Nothing of the source is
stored: only inputs' seeds and the outputs.  tests/test_oracle_golden.py then requires oracle.calli_align.denormalise to
reproduce the outputs bit for bit.

Usage: python scripts/make_golden_tail.py
"""
import ast
import inspect
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs, OUT  # noqa: E402


def reference_tail():
    import InternVL.modeling_internvl_chat as ref_chat
    src = textwrap.dedent(inspect.getsource(ref_chat.InternVLChatModel.calli_align))
    fn = ast.parse(src).body[0]
    body = fn.body
    start = next(i for i, st in enumerate(body)
                 if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and st.targets[0].id == 'outs')
    tail = body[start + 1:]
    assert isinstance(tail[-1], ast.Return)
    args = ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in ('self', 'outs', 'output', 'use_hard_vector_quant', 'drop_zero', 'verbose')],
                         kwonlyargs=[], kw_defaults=[], defaults=[])
    new = ast.FunctionDef(name='calli_align_tail', args=args, body=tail, decorator_list=[], returns=None, type_comments=None)
    mod = ast.fix_missing_locations(ast.Module(body=[new], type_ignores=[]))
    import time
    ns = {'torch': torch, 'time': time}
    exec(compile(mod, '<reference:calli_align tail>', 'exec'), ns)
    first, last = tail[0].lineno, tail[-1].end_lineno
    return ns['calli_align_tail'], (first, last)


sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tail_cases import CASES, make_case  # noqa: E402


def main():
    install_stubs()
    tail, lines = reference_tail()
    gold = {'source_lines': np.array(lines)}
    for name, seed, tiles, vocab, pdt, drop_zero, hard_vq in CASES:
        table, mu, sigma, x, idx, cos = make_case(seed, tiles, vocab, pdt, drop_zero, hard_vq)
        self = types.SimpleNamespace(normed_emb=types.SimpleNamespace(weight=table), mu=mu, sigma=sigma)
        outs = (idx, cos) if hard_vq else idx
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            back, indices = tail(self, outs, x.clone(), hard_vq, drop_zero, False)
        gold[f'{name}.out'] = back.float().numpy()
        gold[f'{name}.out_dtype'] = np.frombuffer(str(back.dtype).encode(), dtype=np.uint8)
        gold[f'{name}.indices'] = indices.numpy()
        print(name, tuple(back.shape), back.dtype)
    path = os.path.join(OUT, 'tail_vectors.npz')
    np.savez_compressed(path, **gold)
    print('wrote', path, os.path.getsize(path), 'bytes; reference lines', lines)


if __name__ == '__main__':
    main()
