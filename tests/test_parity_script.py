"""scripts/real_checkpoint_parity.py: the pure rule behind its exit status, and its command line (the run itself needs a GPU and a checkpoint:
tests/test_gpu_boundary.py drives it on a synthetic checkpoint written to disk)."""
import importlib.util
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location('real_checkpoint_parity', os.path.join(ROOT, 'scripts', 'real_checkpoint_parity.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bf16_step_and_first_divergence():
    m = _load()
    assert m.bf16_step(1.0) == 2.0 ** -7 and m.bf16_step(12.5) == 2.0 ** -4 and m.bf16_step(-0.3) == 2.0 ** -9
    assert m.first_divergence([1, 2, 3], [1, 2, 3]) is None
    assert m.first_divergence([1, 2, 3], [1, 5, 3]) == 1
    assert m.first_divergence([1, 2], [1, 2, 3]) == 2


def test_divergence_is_excused_only_by_a_measured_tie_within_one_bf16_step():
    m = _load()
    ref = torch.zeros(100)
    ref[7], ref[9] = 12.0, 11.9375            # the oracle's gap is exactly one bf16 step at |logit| 8..16 (2^-4)
    hip = ref.clone()
    hip[9] = 12.0625                          # the HIP logits favour id 9 by a measured difference that covers the gap
    r = m.judge_divergence(ref, hip, 7, 9, [], 1.0)
    assert r['excusable'] and abs(r['oracle_gap'] - 0.0625) < 1e-6 and r['one_bf16_step'] == 0.0625
    ref[9] = 11.5                             # a gap of eight steps: nothing excuses a different pick
    assert not m.judge_divergence(ref, hip, 7, 9, [], 1.0)['excusable']
    ref[9] = 11.9375
    assert not m.judge_divergence(ref, ref.clone(), 7, 9, [], 1.0)['excusable']      # identical logits cannot straddle a positive gap
    # the repetition penalty is applied to both sides before the comparison (published 4.45.2 rule: positive scores are divided)
    ref2 = torch.zeros(100)
    ref2[7], ref2[9] = 12.0, 18.0             # raw arg-max 9, but 9 was generated: 18 / 1.5 = 12.0 -> a tie, first index wins
    r = m.judge_divergence(ref2, ref2.clone(), 7, 9, [9], 1.5)
    assert abs(r['oracle_gap']) < 1e-6


def test_command_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'real_checkpoint_parity.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'INTERNVL_PATH' in out.stdout and '--max-new-tokens' in out.stdout
    env = {k: v for k, v in os.environ.items() if k != 'INTERNVL_PATH'}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'real_checkpoint_parity.py')], capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode == 2 and 'checkpoint directory' in bad.stderr
