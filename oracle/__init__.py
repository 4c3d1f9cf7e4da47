"""ORACLE — test infrastructure, not product.

CPU restatement (plain C + numpy/scipy-LAPACK, fp64) of the reference's GPRF block-local
log-likelihood / gradient path (``/root/reference/gprf.py:206-296, 496-591``) and of the callers and
input recipe around it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import this package, and only as the checker.  ``gprf_amd`` never imports it.

Pinning: the reference itself cannot run here (Python 2, treegp / pyublas /
scipy.weave absent), but its published optimisation traces (``gprf_results.tgz``) are exact
known-answer tests; ``tests/test_oracle_kat.py`` reproduces them from seeds to every printed digit
for the ("euclidean","se") kernel.  The ("lld","matern32") kernel is **parity unpinned** (dataset
and treegp absent): it is checked against the in-tree haversine doctests and finite differences only.
"""
