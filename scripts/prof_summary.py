"""Condense rocprofv3 CSV outputs (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def short(name):
    for k in ("k_fill", "k_potrf", "k_solve_panel", "k_solve", "k_at", "k_mtile", "k_gred", "k_gx_finalize", "k_assemble",
              "k_gather_x", "k_gather_y"):
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
