import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~30 s")


@pytest.fixture(scope="session")
def published():
    with open(os.path.join(GOLDEN, "published_traces.json")) as f:
        return json.load(f)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def blocks_from_csr(ptr, pts):
    return [pts[ptr[i]:ptr[i + 1]].astype(np.int64) for i in range(len(ptr) - 1)]


@pytest.fixture(scope="session")
def sdata2000():
    """The reference's n=2000 synthetic run inputs (gprfopt_analyze.py:248-249: lscale=6/sqrt(n),
    obs_std=2/sqrt(n)), sampled once per session by the oracle's recipe."""
    from oracle.harness_ref import SampledDataRef
    ntrain = 2000
    return SampledDataRef(n=ntrain + 500, ntrain=ntrain, lscale=6 / np.sqrt(ntrain), obs_std=2 / np.sqrt(ntrain),
                          yd=50, seed=0)
