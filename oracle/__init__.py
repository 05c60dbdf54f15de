"""TEST INFRASTRUCTURE ONLY — CPU oracle for the DeMo PIFE -> HDM -> ATMoE hot path.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package. The product (`demo2_amd`) never does;
it fails loudly when its HIP library is missing instead of falling back here.
"""
