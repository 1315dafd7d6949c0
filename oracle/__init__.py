"""CPU oracle for the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package restates, on the CPU, the algorithms of the reference's torch-native
attention backend, KV-pool indexing, radix cache and the fp8 / AWQ dequant-GEMM
reference formulas.  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / the reported CPU baseline.  The product
(``ltp-sglang_amd/``) never imports it and has no CPU fallback.

Pinning: each restatement is checked against golden vectors produced by importing the
reference itself in the build container (tests/golden/make_golden.py);
the vectors are committed under tests/golden/*.npz and re-checked by the ``not gpu`` tests.
"""
