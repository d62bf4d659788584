"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU fp32 restatement of the reference's (stockeh/swift) SwinV2
consistency-model forecast path.  It exists so that the HIP path in
``swift_amd/`` can be checked against the reference's arithmetic on a machine
where the reference's Python cannot travel (the GPU box).

Rules (enforced by tests/test_abi_and_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * nothing under ``swift_amd/`` imports it -- the product path fails loudly
    when the HIP extension is missing, it never falls back to this code.

Parity pinning: every function here is checked against golden vectors that
were produced by importing the *actual reference* from ``/root/reference/src``
in the build container (``tools/make_golden.py`` -> ``tests/golden/*.npz``;
the reference has no tests or fixtures of its own).

Each function cites the reference file:line it restates (paths relative to
``/root/reference/src/swift``).
"""
