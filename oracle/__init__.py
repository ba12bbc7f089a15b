"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the GP-reachability hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``safe_exploration_amd/`` imports it;
the product path fails loudly when the HIP library is missing.
"""
