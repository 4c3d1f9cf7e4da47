"""Timeline of the last evaluations in a rocprofv3 rocpd database (kernel-trace): python scripts/trace_timeline.py DB [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cur = db.cursor()
ks = list(cur.execute("select name,start,end,grid_x from kernels order by start"))
t0 = None
prev_end = None
for name, s, e, g in ks[-n:]:
    if t0 is None:
        t0 = s
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    short = name.replace("void gprf::", "").replace("gprf::", "")[:44]
    print("%-44s start %8.1f  gap %6.1f  dur %8.1f  grid %d" % (short, (s - t0) / 1e3, gap, (e - s) / 1e3, g))
    prev_end = e
print("-- averages (us) --")
for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "gprf::" in name:
        print("%-60s calls %5d avg %8.2f" % (name.replace("void gprf::", "").replace("gprf::", "")[:60], calls, avg / 1e3))
