"""CPU oracle for the template-matching path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``enzymm_amd/`` may import this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg.
"""
