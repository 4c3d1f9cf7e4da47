"""Diagnostic: one evaluation with pairs of 17-22 tiles per edge under each Cholesky variant (own interpreter each: the switches
are read once per process) — ll / gradient differences against the default and against the oracle."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DRIVER = r'''
import sys
import numpy as np
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(31)
n = 2400
X = rng.rand(n, 2)
Y = rng.randn(n, 7)
b = Blocker(grid_centers(16))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.09, 0.11], "euclidean", "se"), 0.02, neighbors=b.neighbors())
ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
sz = [len(u) for u in g.block_idxs]
print("pairs", sorted(set((sz[i] + sz[j] + 15) // 16 for i, j in g.neighbors)))
np.savez(sys.argv[1], ll=ll, gX=gX, gC=gC)
g.close()
'''
def main():
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "d.py"), "w").write(DRIVER)
    res = {}
    for tag, env in (("default", {}), ("fill", {"GPRF_DIAG": "fused_fill=0"}), ("reg=0", {"GPRF_DIAG": "potrf_reg=0"}),
                     ("gw=0", {"GPRF_DIAG": "potrf_gw=0"}), ("one queue", {"GPRF_DIAG": "one_queue=1"})):
        e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        e.update(env)
        out = os.path.join(tmp, tag.replace(",", "_").replace("=", "") + ".npz")
        r = subprocess.run([sys.executable, os.path.join(tmp, "d.py"), out], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        print(tag, r.returncode, r.stdout.decode()[-300:].strip())
        if r.returncode == 0:
            res[tag] = np.load(out)
    a = res["default"]
    for tag, b in res.items():
        print("%-14s ll %.15e  dll %.3e  dgX %.3e (rel %.3e)  dgC %.3e" % (tag, float(b["ll"]), float(b["ll"]) - float(a["ll"]),
              np.max(np.abs(b["gX"] - a["gX"])), np.max(np.abs(b["gX"] - a["gX"])) / np.max(np.abs(a["gX"])), np.max(np.abs(b["gC"] - a["gC"]))))
main()
