"""Condense rocprofv3 CSV outputs (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def size_class(name):
    """round 6: the kernels of the by-class pipelines carry their size class as the last template argument"""
    import re
    m = re.search(r"k_solve_panel<\d+, \d+, (?:true|false), \d+, (\d)>", name)
    if m: return {"1": "[large]", "2": "[small]"}.get(m.group(1), "")
    m = re.search(r"k_at_wide<(\d)>", name)
    if m: return {"1": "[large]", "2": "[small]"}.get(m.group(1), "")
    m = re.search(r"k_mgrad<\d, \d, (?:true|false), \d, (?:true|false), (\d)>", name)
    if m: return {"1": "[large]", "2": "[small]"}.get(m.group(1), "")
    return ""


def short(name):
    return short0(name) + size_class(name)


def short0(name):
    # the two instantiations that differ by where K comes from (demangled or mangled spelling)
    if "k_potrf_reg2" in name:
        return "k_potrf_reg2_gen"
    if "k_potrf_reg8w" in name:     # units of 21 .. 28 tiles, waiting tiles in the U pool
        return "k_potrf_reg8w"
    if "k_potrf_reg8" in name:      # k_potrf_reg8<SLOTS, GEN>: K generated, or read from the pool (lld / Matern, forced fills)
        return "k_potrf_reg8_pool" if ("<20, false" in name or "ILi20ELb0E" in name) else "k_potrf_reg8_gen"
    if "k_mgrad" in name and (", true>" in name and "false, 0, true>" in name):
        return "k_mgrad_big"
    if "k_mgrad" in name and ("<0, 0, true>" in name or "ILi0ELi0ELb1E" in name):
        return "k_mgrad_readK"
    for k in ("k_big_gemm", "k_big_diag", "k_big_apply", "k_big_update", "k_big_zz_fold", "k_big_zz", "k_big_init", "k_fill", "k_potrf_reg", "k_potrf", "k_solve_panel", "k_solve", "k_at_wide", "k_at", "k_mgrad", "k_mtile", "k_gred", "k_gx_finalize", "k_assemble",
              "k_assign", "k_route", "k_build_scatter", "k_build", "k_scatter_x", "k_done", "k_finish", "k_pair_max"):
        if k in name:
            return k
    return name[:40]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        nm = short(r.get("Name", ""))
        if nm.startswith("k_"):
            print("%-12s calls %6s  total_ns %12s  avg_ns %10s  min %8s max %8s  pct %s" % (
                nm, r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))

# The kernel-trace pass runs the PRODUCT's launch structure (GPRF_DIAG=tool_env=0: two Cholesky queues forked by the kernel-written
# word and joined by a stream memory operation — rocprofv3 --kernel-trace without counters does not serialise the queues): the
# Cholesky STAGE is the span from the first of its kernels' start to the last one's end, per evaluation, from the trace's own
# timestamps; likewise the whole evaluation (k_assign / k_route start -> k_done end).
import statistics
for f in find("trace", "*kernel_trace.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    evals, cur = [], None
    for r in rows:
        n = short(r.get("Kernel_Name", ""))
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if n in ("k_assign", "k_route"):
            if cur: evals.append(cur)
            cur = {"t0": s_, "potrf": [], "end": None, "q": set()}
        elif cur is not None:
            if n.startswith("k_potrf_reg"):
                cur["potrf"].append((s_, e_))
                cur["q"].add(r.get("Queue_Id"))
            elif n in ("k_done", "k_finish"):
                cur["end"] = e_
    if cur: evals.append(cur)
    two = [c for c in evals if len(c["potrf"]) == 2 and c["end"]]
    if two:
        span = [(max(e for _, e in c["potrf"]) - min(s for s, _ in c["potrf"])) / 1e3 for c in two]
        ovl = [(min(e for _, e in c["potrf"]) - max(s for s, _ in c["potrf"])) / 1e3 for c in two]
        whole = [(c["end"] - c["t0"]) / 1e3 for c in two]
        print("== stage spans from the kernel trace (product launch structure: %d evaluations with both Cholesky kernels, on %d queues) ==" % (
            len(two), max(len(c["q"]) for c in two)))
        print("Cholesky stage (first kernel start -> last kernel end): median %.1f us  min %.1f  max %.1f ; the two kernels overlap %.1f us (median)" % (
            statistics.median(span), min(span), max(span), statistics.median(ovl)))
        print("whole evaluation on the device (k_assign start -> k_done end): median %.1f us  min %.1f  max %.1f" % (
            statistics.median(whole), min(whole), max(whole)))

for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for r in csv.DictReader(open(f)):
            nm = short(r.get("Kernel_Name", ""))
            if not nm.startswith("k_"):
                continue
            c = r.get("Counter_Name")
            agg[nm][c] += float(r.get("Counter_Value", 0))
            cnt[nm][c] += 1
    print("== PMC pass %s: per-launch averages ==" % sub)
    for nm in sorted(agg):
        print("%-12s " % nm + "  ".join("%s=%.4g" % (c, agg[nm][c] / max(cnt[nm][c], 1)) for c in sorted(agg[nm])))


# per-launch HBM traffic of the largest-grid (north-star) launches of each kernel:
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: the counters are in KiB; on gfx950 FETCH_SIZE
# reads exactly half of a wide coalesced stream -> doubled; WRITE_SIZE is exact).  Written as JSON for bench.py.
import json
traffic = {}
vals = defaultdict(lambda: defaultdict(list))
for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            nm = short(r.get("Kernel_Name", ""))
            if nm.startswith("k_") and r.get("Counter_Name") == cname:
                vals[nm][(cname, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
print("== HBM traffic per launch, largest grid of each kernel (KiB counters; read side x2 on gfx950) ==")
for nm in sorted(vals):
    grids = sorted({g for (_, g) in vals[nm]})
    g = grids[-1]
    fe = vals[nm].get(("FETCH_SIZE", g), [0.0]); wr = vals[nm].get(("WRITE_SIZE", g), [0.0])
    fetch_kib, write_kib = sum(fe) / len(fe), sum(wr) / len(wr)
    traffic[nm] = {"grid": g, "fetch_KiB": fetch_kib, "write_KiB": write_kib,
                   "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0}
    print("%-14s grid %8d  FETCH_SIZE %10.0f KiB  WRITE_SIZE %10.0f KiB  -> %.1f MB" % (nm, g, fetch_kib, write_kib, traffic[nm]["bytes_per_launch"] / 1e6))
# the native sources these counters belong to (bench.py reports the traffic only for exactly these)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gprf_amd import build as hip_build
traffic["source_hash"] = hip_build.source_hash()
with open(os.path.join(out, "traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1, sort_keys=True)
