"""TEST INFRASTRUCTURE ONLY — CPU oracle for the doubly-stochastic DGP hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``doubly-stochastic-dgp_amd/``) never imports this package and fails loudly when the
HIP library is missing.

PARITY UNPINNED (numerically): the reference (GPflow 1.1.1 / TensorFlow 1.8) cannot be imported in this
container and holds no golden vectors; the oracle is pinned *relationally* by re-stating the identities
the reference's own tests assert (tests/test_dgp.py, tests/test_collapsed.py) against independent closed
forms — see ``tests/test_oracle_identities.py`` and DESIGN.md §3.
"""
