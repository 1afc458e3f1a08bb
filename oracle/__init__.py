"""CPU oracle for the SSL4GIE hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in ``ssl4gie_amd`` (the product) may import from this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
