"""CPU oracle for the SmoQyElPhQMC hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package; the product package never does.  Parity status: "parity unpinned" (see
the header of ``smoqy_oracle.c``).
"""
