"""Does a more accurate row panel in EVERY unit make the ASSEMBLED Bethe gradient more accurate?  On MI355X it did not
(round 5: one step of refinement on the Cholesky's row panels took the per-unit error ratio against LAPACK from 0.89 to 0.86 and
the assembled north-star gradient from 0.78x LAPACK's error to 1.07x; refining only the units of at most 11 tiles: 1.34x).  The
assembled gradient of a point of block i is  sum_j g_(i,j) - (deg_i - 1) g_i : eight pair terms and -7 times the unary term, whose
rounding errors are NOT independent (a pair's factorisation starts with the very arithmetic of its first block's).  This script
emulates that on the north-star data, centre blocks with all eight neighbours: for each panel variant of
tests/diag/cpu_row_panel_emulation.py the error of the assembled gradient rows of the centre block against the 80-bit
evaluation, next to the per-unit errors.  CPU only:  python tests/diag/cpu_bethe_cancellation.py [n_centres]"""
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ld_truth import unit_llgrad_ld
from cpu_row_panel_emulation import chol_blocked, M_from_U, grad_from

MODES = ("subst", "vinv", "vinv_ref", "mixed11")


def main():
    from gprf_amd.synthetic import SampledData
    from gprf_amd import grid_centers
    n_centres = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sd = SampledData(n=10500, ntrain=10000, lscale=0.06, obs_std=0.02, yd=50, seed=0, use_gpu=False,
                     cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gprf_bench_data"))
    sd.set_centers(grid_centers(100))
    ls = np.array([0.06, 0.06])
    deg = np.zeros(100, dtype=int)
    for (i, j) in sd.neighbors:
        deg[i] += 1; deg[j] += 1
    rng = np.random.RandomState(7)
    centres = rng.choice(np.nonzero(deg == 8)[0], n_centres, replace=False)
    tot = {k: [] for k in ("lapack",) + MODES}
    for c in centres:
        units = [(sd.block_idxs[c], 1 - deg[c], slice(0, len(sd.block_idxs[c])))]
        for (i, j) in sd.neighbors:
            if c in (i, j):
                idx = np.concatenate([sd.block_idxs[i], sd.block_idxs[j]])
                ni = len(sd.block_idxs[i])
                units.append((idx, 1, slice(0, ni) if c == i else slice(ni, len(idx))))
        acc = {k: 0.0 for k in ("truth", "lapack") + MODES}
        per_unit = {k: [] for k in ("lapack",) + MODES}
        for idx, w, rows in units:
            Xu, Yu = sd.X_obs[idx], sd.SY[idx]
            d = (Xu[:, None, :] - Xu[None, :, :]) / ls
            Knf = np.exp(-np.sum(d * d, axis=2)); K = Knf + 0.01 * np.eye(len(idx))
            _, gt = unit_llgrad_ld(Xu, Yu, 0.01, 1.0, ls)
            acc["truth"] = acc["truth"] + w * gt[rows]
            T = (len(idx) + 15) // 16
            Us = {"lapack": sl.cholesky(K, lower=False)}
            for md in MODES:
                Us[md] = chol_blocked(K, ("vinv_ref" if T <= 11 else "vinv") if md == "mixed11" else md)
            for k, U in Us.items():
                g = grad_from(M_from_U(U, Yu, 50), Xu, Knf, ls)
                acc[k] = acc[k] + w * g[rows]
                per_unit[k].append(np.max(np.abs(g - gt.astype(np.float64))))
        line = {}
        for k in ("lapack",) + MODES:
            e = float(np.max(np.abs(acc[k] - acc["truth"])))
            tot[k].append(e)
            line[k] = "%.2e (units %.2e)" % (e, np.mean(per_unit[k]))
        print("centre block", c, len(units), "units:", line, flush=True)
    print("assembled-gradient error of the centre blocks' rows, mean over", len(centres), "centres; ratio to LAPACK")
    for k, v in tot.items():
        print("  %-9s %.3e   %.2f" % (k, np.mean(v), np.mean(v) / np.mean(tot["lapack"])))


if __name__ == "__main__":
    main()
