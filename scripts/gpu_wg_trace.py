"""Where and when every workgroup of one kernel ran (diagnostic build):
   GPRF_LIB=$PWD/gprf_amd/libgprf_trace.so GPRF_BUILD_DEFS="-DGPRF_WGTRACE=1" python gprf_amd/build.py   (1 solve, 2 at, 3 mgrad, 4 / 5 Cholesky big / small)
   GPRF_LIB=... KERNEL=1 python scripts/gpu_wg_trace.py
Prints: kernel span, workgroup durations by tag, resident workgroups per CU over time (mean / histogram), per-XCD finish
times — the numbers behind DESIGN section 4's "what bounds the GEMM-shaped stages"."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gprf_amd import GPCov, Blocker, grid_centers
from gprf_amd.gprf import GPRF
from gprf_amd import _capi
rng = np.random.RandomState(1)
n = 10000; X = rng.rand(n, 2); Y = rng.randn(n, 50)
b = Blocker(grid_centers(100)); blocks = b.block_clusters(X); nbrs = b.neighbors()
g = GPRF(X, Y, None, GPCov([1.0], [0.06, 0.06], "euclidean", "se"), 0.01, block_idxs=blocks, neighbors=nbrs)
g._push_neighbors(nbrs)
ctx = g._ctx
kern = int(os.environ.get("KERNEL", "1"))
stop = {1: 3, 2: 4, 3: 6, 4: 2, 5: 2}[kern]
for _ in range(3): ctx.debug_run(X, stop)
out = np.zeros(4 * (1 << 15))
ctx._check(ctx.lib.gprf_debug_fetch(ctx.h, 0, 11, _capi.dptr(out), out.size), "fetch")
rec = out.reshape(-1, 4)
rec = rec[rec[:, 1] > 0]
t0, t1, hw, tag = rec[:, 0], rec[:, 1], rec[:, 2].astype(np.int64), rec[:, 3].astype(np.int64)
if kern in (4, 5):          # the Cholesky kernels pack (ticks before the step loop, ticks inside it) behind the tag
    loop_ticks = tag // 10000000
    pro_ticks = (tag % 10000000) // 1000
    tag = tag % 1000
    for tg in sorted(set(tag.tolist())):
        sel = tag == tg
        if sel.sum() >= 20:
            print("  tag %4d: before the step loop %.0f ticks, the loop %.0f, after it %.0f (means, 10 ns ticks)"
                  % (tg, pro_ticks[sel].mean(), loop_ticks[sel].mean(), ((t1 - t0)[sel] - pro_ticks[sel] - loop_ticks[sel]).mean()))
xcc = hw >> 32
if os.environ.get("PER_XCC"):          # counters that are not synchronised across the XCDs: align their first starts
    for x in set(xcc.tolist()):
        b_ = t0[xcc == x].min(); t0[xcc == x] -= b_; t1[xcc == x] -= b_
base = t0.min()
t0 = t0 - base; t1 = t1 - base
span = t1.max()
tick_ns = float(os.environ.get("TICK_NS", "10"))      # s_memtime: 100 MHz constant-rate counter
print("workgroups that did work: %d   span %.0f ticks = %.1f us" % (len(rec), span, span * tick_ns / 1e3))
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct CUs seen: %d" % len(set(cuid.tolist())))
dur = t1 - t0
for tg in sorted(set(tag.tolist())):
    sel = tag == tg
    if sel.sum() >= 20:
        print("  tag %5d: n=%4d  duration mean %.0f  min %.0f  max %.0f ticks   start mean %.0f  end max %.0f"
              % (tg, sel.sum(), dur[sel].mean(), dur[sel].min(), dur[sel].max(), t0[sel].mean(), t1[sel].max()))
# resident workgroups over time
grid = np.linspace(0, span, 41)
print("time(ticks) -> resident workgroups (all CUs) / CUs with at least one")
for a in grid[:-1]:
    live = (t0 <= a) & (t1 > a)
    print("  %6.0f  %5d  %4d" % (a, live.sum(), len(set(cuid[live].tolist()))))
print("sum of workgroup durations / span = %.1f resident on average" % (dur.sum() / span))
for x in sorted(set(xcc.tolist())):
    print("  xcc %d: %d workgroups, last end %.0f" % (x, (xcc == x).sum(), t1[xcc == x].max()))
g.close()
