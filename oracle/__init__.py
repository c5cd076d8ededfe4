"""CPU oracle for the pylrbms hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain NumPy/SciPy fp64 restatement of the per-subdomain
offline/online path of dune-community/pylrbms (block SWIPDG assembly, Oswald
interpolation error, RT0 diffusive-flux reconstruction, estimator products,
Galerkin projection, localized a-posteriori estimator, reduced solve).

PARITY UNPINNED: the arithmetic of the reference lives in dune-gdt / dune-xt and
a private pyMOR fork, none of which is vendored in the reference tree or
installed here, and the reference's own tests hold no numerical fixture for
this path (python/test/base.py:13-15, python/test/mpitest.py:11-46,
dune/pylrbms/test/empty.cc:31).  The oracle therefore follows the *structure*
of the reference files it cites and the published SWIPDG / OS2015 estimator
mathematics; it is pinned only by structural invariants (tests/test_oracle_*.py)
and the soft known-answer values printed by
python/scripts/linearelliptic_block_swipdg_decomp.py:41-43.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (pylrbms_amd) never does.
"""
