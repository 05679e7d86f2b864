"""CPU oracle for the LiveTalking lip-sync render hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.  The product
path (``livetalking_amd``) never imports this package and fails loudly when the
HIP library is missing.

Every function cites the reference file:line it restates (paths relative to
the upstream LiveTalking checkout).  Pinning status is stated per module.
"""
