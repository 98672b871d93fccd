"""CPU oracle for the LongSpec draft-then-verify hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import, call, link or execute it, and there only as the
checker -- never as the thing that is measured or shipped.  The product path
(``longspec_amd``) calls hand-written HIP kernels through the C-ABI library and
fails loudly when that library is missing; it never falls back to this code.
"""
