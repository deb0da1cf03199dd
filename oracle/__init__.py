"""CPU oracle for dgNN's fused operators -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``dgnn_amd``) never does.
See ``dgnn_oracle.c`` for the reference citations and the parity status
("parity unpinned" against reference outputs; formula-pinned).
"""
from .oracle import *  # noqa: F401,F403
