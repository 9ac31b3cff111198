"""Seeded inputs of the calli_align-tail vectors: shared by scripts/make_golden_tail.py (which runs the reference's own
statements on them) and tests/test_oracle_golden.py (which runs the oracle on them).  Data only."""
import torch

D = 256      # the tail is element-wise: the width is irrelevant to what is pinned here (GPU parity at 4096: tests/test_gpu_calli.py)

CASES = [  # (name, seed, tiles, vocab, param dtype, drop_zero, hard_vq)
    ('fp32_plain', 1, 5, 512, torch.float32, False, False),
    ('bf16_plain', 2, 5, 512, torch.bfloat16, False, False),
    ('fp32_drop_zero', 3, 6, 512, torch.float32, True, False),
    ('bf16_hard_vq', 4, 4, 512, torch.bfloat16, False, True),
    ('fp32_hard_vq_drop_zero', 5, 6, 512, torch.float32, True, True),
    ('single_tile', 6, 1, 512, torch.bfloat16, False, False),
]


def make_case(seed, n_tiles, vocab, param_dtype, plant_zero, low_cos):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(vocab, D, generator=g).to(torch.bfloat16)
    mu = (torch.randn(vocab, 1, generator=g) * 0.002).to(param_dtype)
    sigma = (0.02 + torch.randn(vocab, 1, generator=g) * 0.002).to(param_dtype)
    x = torch.randn(n_tiles, 3, D, generator=g).to(torch.bfloat16)
    idx = torch.randint(1, vocab, (n_tiles, 3), generator=g)
    cos = (0.55 + 0.4 * torch.rand(n_tiles, 3, generator=g)).to(torch.bfloat16)
    if plant_zero:
        idx[0, 1] = 0
        idx[-1, 2] = 0
    if low_cos:
        cos[0, 0] = 0.5            # boundary: <= thresh replaces
        cos[1, 1] = 0.25
    return table, mu, sigma, x, idx, cos
