"""INTEGRATION.md's stub B — the code a maintainer of the reference pastes into gprf.py — executed VERBATIM: the python
block is cut out of the document and run in a fresh interpreter that imports neither gprf_amd nor torch (raw
ctypes.CDLL on libgprf_hip.so), on an object with exactly the attributes the reference's GPRF holds, against the golden
vector tests/golden/c1_small.npz (BASELINE configs[0]: 500 points, 4 blocks, 6 pairs, yd = 10)."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DRIVER = r'''
import collections, sys, numpy as np
exec(compile(open(sys.argv[1]).read(), "INTEGRATION.md:stub-B", "exec"))
assert "torch" not in sys.modules and "gprf_amd" not in sys.modules
z = np.load(sys.argv[2])
GPCov = collections.namedtuple("GPCov", "wfn_params dfn_params dfn_str wfn_str")     # treegp.gp.GPCov's fields

class GPRF(HipLLGrad, object):                    # what is left of the reference's class around the stub
    def __init__(self, X, Y, cov, noise_var, block_idxs, neighbors):
        self.X, self.Y, self.cov, self.noise_var = X, Y, cov, noise_var
        self.block_idxs, self.n_blocks, self.neighbors = block_idxs, len(block_idxs), neighbors
        self._hip_init()

th, ptr, pts = z["theta"], z["block_ptr"], z["block_pts"]
blocks = [pts[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]
g = GPRF(z["X_obs"], z["SY"], GPCov([th[1]], th[2:], "euclidean", "se"), th[0], blocks, [tuple(r) for r in z["neighbors"]])
ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
assert abs(ll - float(z["ll_gprf"])) <= 1e-12 * abs(float(z["ll_gprf"])), (ll, float(z["ll_gprf"]))
assert np.max(np.abs(gX - z["gX_gprf"])) <= 1e-9 * np.max(np.abs(z["gX_gprf"]))
assert np.allclose(gC, z["gC_gprf"], rtol=1e-9)
ll2, gX2, gC2 = g.llgrad(local=False, grad_X=True)
assert abs(ll2 - float(z["ll_allpairs"])) <= 1e-12 * abs(float(z["ll_allpairs"])) and gC2.shape == (0, 0)
assert np.max(np.abs(gX2 - z["gX_allpairs"])) <= 1e-9 * np.max(np.abs(z["gX_allpairs"]))
g.neighbors = []
ll3 = g.llgrad()[0]
assert abs(ll3 - float(z["ll_local"])) <= 1e-12 * abs(float(z["ll_local"]))
g._hip_close()
print("stub B ok", ll, ll2, ll3)
'''


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sect = text[text.index("## B. Minimal stub"):]
    return re.search(r"```python\n(.*?)```", sect, flags=re.S).group(1)


def test_stub_b_runs_verbatim(tmp_path):
    from gprf_amd import build
    (tmp_path / "stub_b.py").write_text(stub_source())
    (tmp_path / "driver.py").write_text(DRIVER)
    env = dict(os.environ, GPRF_HIP_LIB=build.LIB)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(tmp_path / "driver.py"), str(tmp_path / "stub_b.py"),
                        os.path.join(ROOT, "tests", "golden", "c1_small.npz")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "stub B ok" in r.stdout
