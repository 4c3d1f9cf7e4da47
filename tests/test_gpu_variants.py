"""The launch variants of one evaluation agree bit for bit: the single-launch table build (k_build_scatter) against
the three-launch one (k_build x 2 + k_scatter_x), the gradient partials folded inside the assembly against a
k_gx_finalize launch, the Cholesky as two instantiations on two queues against one queue.  The forms are selected through the
library's ONE diagnostic switch (GPRF_DIAG="key=value,..."); every variant runs in its own interpreter on the same seeded
walk — points cross block borders at every step, so the tables are rebuilt on the device each time — and prints a
digest of everything the walk returned."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DRIVER = r'''
import hashlib, sys
import numpy as np
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF

import os
rng = np.random.RandomState(31)
n = int(os.environ.get("VAR_N", "1800"))          # 16 blocks of ~112 points: pairs of 13-15 tiles per edge (both Cholesky classes)
X = rng.rand(n, 2)
Y = rng.randn(n, 7)
b = Blocker(grid_centers(int(os.environ.get("VAR_BLOCKS", "16"))))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.09, 0.11], "euclidean", "se"), 0.02, neighbors=b.neighbors())
h = hashlib.sha256()
moved = 0
for it in range(6):
    Xn = np.clip(X + 0.03 * rng.randn(n, 2), 0.0, 1.0)
    before = [len(u) for u in g.block_idxs]
    g.update_X(Xn)
    ll, gX, gC = g.llgrad(grad_X=True, grad_cov=(it % 2 == 0))
    moved += before != [len(u) for u in g.block_idxs]
    h.update(np.float64(ll).tobytes()); h.update(np.ascontiguousarray(gX).tobytes()); h.update(np.ascontiguousarray(gC).tobytes())
    for u in g.block_idxs:
        h.update(np.ascontiguousarray(u, dtype=np.int64).tobytes())
    X = Xn
assert moved >= 4, moved
g.close()
print("DIGEST", h.hexdigest())
'''


def run_variant(tmp_path, extra_env):
    (tmp_path / "driver.py").write_text(DRIVER)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.update(extra_env)
    r = subprocess.run([sys.executable, str(tmp_path / "driver.py")], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("DIGEST ")]
    assert len(lines) == 1, out[-3000:]
    return lines[0].split()[1]


def test_single_launch_table_build_with_many_blocks(tmp_path):
    """400 blocks (more than one per thread of a workgroup), 1900 units, 3400 CSR entries: inside k_build_scatter's limits,
    beyond the assembly's fold (k_gx_finalize runs)"""
    shape = {"VAR_N": "4000", "VAR_BLOCKS": "400"}
    base = run_variant(tmp_path, shape)
    assert run_variant(tmp_path, dict(shape, GPRF_DIAG="fused_build=0")) == base
    # a launch more than two rounds of CUs deep: the gradient grid is walked part by part in groups of 64 launch slots
    assert run_variant(tmp_path, dict(shape, GPRF_DIAG="part_major=0")) == base
    # solve -> At -> gradient as two pipelines side by side (off by default: measured no faster) against launch-wide stages
    assert run_variant(tmp_path, dict(shape, GPRF_DIAG="pipe=50")) == base
    assert run_variant(tmp_path, dict(shape, GPRF_DIAG="pipe=20")) == base


def test_pipelined_stages_equal_launch_wide_stages_bit_for_bit(tmp_path):
    """The north-star shape's unit count (100 blocks + 342 pairs): the split solve -> At -> gradient launches — main queue /
    low-priority queue, cut at 50 %, 25 % and 75 % of the launch order — against the launch-wide form (the default: the split
    was measured no faster, profiles/r05_pipeline_ab.txt) and against it with one Cholesky queue and the unit-by-unit walk."""
    shape = {"VAR_N": "5000", "VAR_BLOCKS": "100"}
    base = run_variant(tmp_path, shape)
    for d in ("pipe=50", "pipe=25", "pipe=75", "one_queue=1,part_major=0", "solve_class=0", "class_depth=2"):
        assert run_variant(tmp_path, dict(shape, GPRF_DIAG=d)) == base, d


@pytest.mark.parametrize("n,blocks", [(3700, 36), (1600, 16), (2100, 16)])
def test_stages_pipelined_by_size_class_equal_launch_wide_stages_bit_for_bit(tmp_path, n, blocks):
    """Round 6: in a two-queue launch each size class's substitution, At and gradient follow that class's Cholesky kernel on its
    queue (the small class's substitution as a 13-tile instantiation at four workgroups per CU; the large class's as the 16- or
    the 20-tile one).  Shapes: 36 blocks with pairs of 11-15 tiles (the north star's mix), 16 blocks with pairs of 12-15, and
    pairs of 15-19 (the 20-tile large-class instantiation).  solve_class=0 runs every stage as one launch behind the join;
    class_depth=1 / 2 stop the pipelines after the substitution / after At: every result of the walk the same bits"""
    shape = {"VAR_N": str(n), "VAR_BLOCKS": str(blocks)}
    base = run_variant(tmp_path, shape)
    for d in ("solve_class=0", "class_depth=1", "class_depth=2", "tail_swap=0"):
        assert run_variant(tmp_path, dict(shape, GPRF_DIAG=d)) == base, d


DX3_DRIVER = r'''
import sys
import numpy as np
from gprf_amd import GPCov
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(3)
n = 1000
X = rng.rand(n, 3)
Y = rng.randn(n, 6)
order = np.argsort(X[:, 0], kind="stable")
blocks = [np.sort(order[k * 125:(k + 1) * 125]) for k in range(8)]        # eight slabs of 125 points: pairs of 250 = 16 tiles
nbrs = [(k + 1, k) for k in range(7)]
g = GPRF(X, Y, None, GPCov([1.0], [0.3, 0.25, 0.35], "euclidean", "se"), 0.02, block_idxs=blocks, neighbors=nbrs)
ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
np.savez(sys.argv[1], ll=ll, gX=gX, gC=gC)
g.close()
'''


def test_three_dimensional_inputs_through_the_by_class_pipelines(tmp_path):
    """dx = 3: the gradient kernel's general instantiation has no class form, so the pipelines stop behind At (class_depth is
    capped at 2) and the gradient runs launch-wide behind the join — against every stage launch-wide, bit for bit"""
    import numpy as np
    (tmp_path / "dx3.py").write_text(DX3_DRIVER)
    out = {}
    for tag, env in (("by class", {}), ("launch wide", {"GPRF_DIAG": "solve_class=0"})):
        e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        e.update(env)
        f = str(tmp_path / (tag.replace(" ", "_") + ".npz"))
        r = subprocess.run([sys.executable, str(tmp_path / "dx3.py"), f], cwd=str(tmp_path), env=e, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        out[tag] = np.load(f)
    a, b = out["by class"], out["launch wide"]
    assert float(a["ll"]) == float(b["ll"]) and np.array_equal(a["gX"], b["gX"]) and np.array_equal(a["gC"], b["gC"])


def test_launch_variants_agree_bit_for_bit(tmp_path):
    base = run_variant(tmp_path, {})
    for name, env in (("three-launch table build", {"GPRF_DIAG": "fused_build=0"}),
                      ("k_gx_finalize as a launch", {"GPRF_DIAG": "gx_fold=0"}),
                      ("one Cholesky queue", {"GPRF_DIAG": "one_queue=1"}),
                      ("fork / join of the two Cholesky queues by events", {"GPRF_DIAG": "side_events=1"}),
                      ("solve / gradient grids walked unit by unit", {"GPRF_DIAG": "part_major=0"}),
                      ("nearest centre by the full scan instead of the grid's 3 x 3", {"GPRF_DIAG": "grid_hint=0"}),
                      # round 6: each size class's forward substitution behind its own Cholesky kernel on that kernel's queue (the
                      # small class with a 13-tile instantiation at four workgroups per CU) against ONE launch behind the join
                      ("the substitution as one launch", {"GPRF_DIAG": "solve_class=0"}),
                      ("the queues joined into the main queue instead of the side queue", {"GPRF_DIAG": "tail_swap=0"}),
                      ("only the substitution by class", {"GPRF_DIAG": "class_depth=1"}),
                      ("substitution and At by class", {"GPRF_DIAG": "class_depth=2"}),
                      ("all of these at once", {"GPRF_DIAG": "fused_build=0,gx_fold=0,one_queue=1,part_major=0"})):
        assert run_variant(tmp_path, env) == base, name


WIDE_DRIVER = r'''
import sys
import numpy as np
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
import os
rng = np.random.RandomState(31)
n = int(os.environ.get("WIDE_N", "2400"))
X = rng.rand(n, 2)
Y = rng.randn(n, 7)
b = Blocker(grid_centers(16))
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.09, 0.11], "euclidean", "se"), 0.02, neighbors=b.neighbors())
sz = [len(u) for u in g.block_idxs]
tiles = sorted(set((sz[i] + sz[j] + 15) // 16 for i, j in g.neighbors))
if n == 2400:
    assert tiles[0] >= 17 and 19 in tiles and 20 in tiles and tiles[-1] > 20, tiles
elif n == 3900:
    assert 31 in tiles and 32 in tiles and tiles[0] >= 26, tiles
else:
    assert tiles[0] >= 21 and tiles[-1] > 28 and len([t for t in tiles if 21 <= t <= 28]) >= 5, tiles
ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
np.savez(sys.argv[1], ll=ll, gX=gX, gC=gC)
g.close()
'''


def test_units_of_17_to_20_tiles_on_the_eight_wave_cholesky(tmp_path):
    """16 blocks of ~150 points: pairs of 17-22 tiles per edge.  The eight-wave register kernel takes those of up to 20 (160
    tiles in accumulators, up to 30 waiting in LDS), generating K (default) or reading it from the pool (GPRF_DIAG fused_fill=0).
    potrf_reg=0 sends every unit through the K pool and the generic kernel: the same arithmetic per tile in the same order (row panel V_jj^T C_jk on the matrix pipe,
    the step's products from zero and one addition), the same bits"""
    import numpy as np
    (tmp_path / "wide.py").write_text(WIDE_DRIVER)
    out = {}
    for tag, env in (("gen", {}), ("pool", {"GPRF_DIAG": "fused_fill=0"}), ("queue", {"GPRF_DIAG": "one_queue=1"}),
                     ("all generic", {"GPRF_DIAG": "potrf_reg=0"})):
        e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        e.update(env)
        r = subprocess.run([sys.executable, str(tmp_path / "wide.py"), str(tmp_path / (tag + ".npz"))], cwd=str(tmp_path), env=e,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        out[tag] = np.load(str(tmp_path / (tag + ".npz")))
    a = out["gen"]
    for tag in ("pool", "queue", "all generic"):
        b = out[tag]
        assert float(a["ll"]) == float(b["ll"]) and np.array_equal(a["gX"], b["gX"]) and np.array_equal(a["gC"], b["gC"]), tag


def test_units_of_21_to_32_tiles_wait_in_the_U_pool(tmp_path):
    """16 blocks of ~206 points: pairs of 24-30 tiles per edge.  A launch with units above 20 tiles goes through the K pool as a
    whole and ONE eight-wave register kernel takes every unit of up to 32 tiles (28 until round 5), the tiles beyond its 160
    accumulator slots waiting in the U pool (in place, through L2).  GPRF_DIAG potrf_gw=0: the generating kernels for units of
    up to 20 tiles, the generic kernel above — the same bits.  (31- and 32-tile units: the next test.)"""
    import numpy as np
    (tmp_path / "wide.py").write_text(WIDE_DRIVER)
    out = {}
    for tag, env in (("gw", {}), ("pool", {"GPRF_DIAG": "fused_fill=0"}), ("generic", {"GPRF_DIAG": "potrf_gw=0"}),
                     ("pool generic", {"GPRF_DIAG": "fused_fill=0,potrf_gw=0"}), ("all generic", {"GPRF_DIAG": "potrf_reg=0"})):
        e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), WIDE_N="3300")
        e.update(env)
        r = subprocess.run([sys.executable, str(tmp_path / "wide.py"), str(tmp_path / (tag.replace(" ", "_") + ".npz"))], cwd=str(tmp_path),
                           env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        out[tag] = np.load(str(tmp_path / (tag.replace(" ", "_") + ".npz")))
    a = out["gw"]
    for tag in ("pool", "generic", "pool generic", "all generic"):
        b = out[tag]
        assert float(a["ll"]) == float(b["ll"]) and np.array_equal(a["gX"], b["gX"]) and np.array_equal(a["gC"], b["gC"]), tag


def test_units_of_31_and_32_tiles_against_the_generic_kernel(tmp_path):
    """16 blocks of ~244 points: pairs of 26-34 tiles per edge, among them 31 and 32 (496 .. 512 points: the last sizes the
    eight-wave kernel takes, up to 336 tiles waiting in the U pool, 141 KB of LDS), the larger ones on the blocked path beside
    them.  Against potrf_gw=0 (generic kernel above 20 tiles) and potrf_reg=0 (generic kernel for everything): the same bits"""
    import numpy as np
    (tmp_path / "wide.py").write_text(WIDE_DRIVER)
    out = {}
    for tag, env in (("gw", {}), ("generic", {"GPRF_DIAG": "potrf_gw=0"}), ("all generic", {"GPRF_DIAG": "potrf_reg=0"})):
        e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), WIDE_N="3900")
        e.update(env)
        r = subprocess.run([sys.executable, str(tmp_path / "wide.py"), str(tmp_path / (tag.replace(" ", "_") + ".npz"))], cwd=str(tmp_path),
                           env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        out[tag] = np.load(str(tmp_path / (tag.replace(" ", "_") + ".npz")))
    a = out["gw"]
    for tag in ("generic", "all generic"):
        b = out[tag]
        assert float(a["ll"]) == float(b["ll"]) and np.array_equal(a["gX"], b["gX"]) and np.array_equal(a["gC"], b["gC"]), tag


def test_blocked_path_beside_the_one_workgroup_kernels_bit_for_bit(tmp_path):
    """A launch with units on BOTH sides of the one-workgroup limit (25 blocks of ~400 points = 25 tiles: the eight-wave
    kernel; their 72 pairs of ~800 points: the blocked path): by default the blocked Cholesky / substitution run BESIDE the
    one-workgroup kernels on a third queue, joined by stream memory operations.  Against one after the other
    (big_beside=0), and that with one Cholesky queue: a cross-queue race would show as different bits on a walk of six
    re-partitioned evaluations"""
    shape = {"VAR_N": "10000", "VAR_BLOCKS": "25"}
    base = run_variant(tmp_path, shape)
    for d in ("big_beside=0", "one_queue=1,big_beside=0"):
        assert run_variant(tmp_path, dict(shape, GPRF_DIAG=d)) == base, d


FEW_WIDE_DRIVER = r'''
import sys
import numpy as np
from gprf_amd import Blocker, grid_centers, GPCov
from gprf_amd.gprf import GPRF
rng = np.random.RandomState(5)
n = 2200
X = rng.rand(n, 2)
c = np.array(grid_centers(25))
X[:230] = c[12] + 0.06 * (rng.rand(230, 2) - 0.5)          # the centre block crowded: its eight pairs exceed 320 points
Y = rng.randn(n, 5)
b = Blocker(c)
g = GPRF(X, Y, b.block_clusters, GPCov([1.0], [0.09, 0.11], "euclidean", "se"), 0.02, neighbors=b.neighbors())
sz = [len(u) for u in g.block_idxs]
tiles = [(sz[i] + sz[j] + 15) // 16 for i, j in g.neighbors]
wide = [t for t in tiles if t > 20]
assert 4 <= len(wide) < 16 and max(wide) <= 28 and min(tiles) <= 13, sorted(tiles)
ll, gX, gC = g.llgrad(grad_X=True, grad_cov=True)
np.savez(sys.argv[1], ll=ll, gX=gX, gC=gC)
g.close()
'''


def test_a_few_wide_units_leave_the_others_generated(tmp_path):
    """A north-star-shaped partition with ONE crowded block: its eight pairs have 21-28 tiles per edge, everything else at most
    13.  The launch stays a generating one (round 2's per-unit decision): the wide pairs are filled and take the eight-wave
    kernel with waiting tiles in the U pool BEHIND the generating kernels (GPRF_DIAG potrf_gw=0: the generic kernel) — the same
    bits as everything through the pool"""
    import numpy as np
    (tmp_path / "few.py").write_text(FEW_WIDE_DRIVER)
    out = {}
    for tag, env in (("gen", {}), ("generic", {"GPRF_DIAG": "potrf_gw=0"}), ("pool", {"GPRF_DIAG": "fused_fill=0"})):
        e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        e.update(env)
        r = subprocess.run([sys.executable, str(tmp_path / "few.py"), str(tmp_path / (tag + ".npz"))], cwd=str(tmp_path), env=e,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        out[tag] = np.load(str(tmp_path / (tag + ".npz")))
    a = out["gen"]
    for tag in ("generic", "pool"):
        b = out[tag]
        assert float(a["ll"]) == float(b["ll"]) and np.array_equal(a["gX"], b["gX"]) and np.array_equal(a["gC"], b["gC"]), tag


def test_generated_K_equals_filled_K_bit_for_bit(tmp_path):
    """K generated inside the register Cholesky = K filled into the pool (k_fill_se, GPRF_DIAG fused_fill=0) and read, entry
    for entry (both add the noise to the rounded kernel value: two roundings, as the reference does): the whole walk comes
    out the same"""
    assert run_variant(tmp_path, {"GPRF_DIAG": "fused_fill=0"}) == run_variant(tmp_path, {})
