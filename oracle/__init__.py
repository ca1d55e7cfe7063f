"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference (tensorboy/centerpose) hot path used as the parity
checker.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything under ``oracle/``.  The product package ``centerpose_amd`` never
imports it and has no CPU fallback: it fails loudly when the HIP library is missing.
"""
