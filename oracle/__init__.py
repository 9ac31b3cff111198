"""CPU oracle for the CalliReader image->text hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, eager) restatement of the reference's
algorithm for the path in SURVEY.md section 8(a).  It exists to CHECK the HIP
path; it is never the product.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  `callireader_amd/` must not.

Pinning status: the reference ships no tests and no golden
outputs for this path, so the oracle is pinned against outputs of the
reference's own modules, imported from /root/reference in the build container
with seeded random weights (`scripts/make_golden.py` -> `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).  One piece cannot be pinned that
way: the greedy generation loop lives in transformers==4.45.2
(`GenerationMixin._sample`, `RepetitionPenaltyLogitsProcessor`), which is not
installed here and cannot run against the installed transformers 5.x
(SURVEY.md 8c).  `oracle/generate.py` restates its published semantics and is
anchored on the reference's call sites; for that loop: PARITY UNPINNED.
"""
