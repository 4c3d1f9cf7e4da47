"""Covariance record with the fields the reference reads from ``treegp.gp.GPCov``
(gprf.py:109,127,163,364-371; gprfopt.py:64, synthetic.py:149, run_seismic.py:299-301)."""
from collections import namedtuple

GPCov = namedtuple("GPCov", ["wfn_params", "dfn_params", "dfn_str", "wfn_str"])

SUPPORTED = {("euclidean", "se"), ("lld", "matern32")}
