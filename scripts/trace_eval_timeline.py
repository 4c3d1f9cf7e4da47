"""One evaluation's kernel timeline from a rocprofv3 --kernel-trace CSV (product launch structure: GPRF_DIAG=tool_env=0):
   python scripts/trace_eval_timeline.py <dir with *kernel_trace.csv> [evaluation index]"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
evals, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "k_assign" in n or "k_route" in n:
        if cur: evals.append(cur)
        cur = []
    if cur is not None:
        cur.append(r)
if cur: evals.append(cur)
ev = evals[which]
t0 = int(ev[0]["Start_Timestamp"])
qs = sorted({r["Queue_Id"] for r in ev})
print("evaluation %d of %d; queues %s" % (which % len(evals), len(evals), qs))
for r in ev:
    n = r["Kernel_Name"].replace("void gprf::", "").replace("gprf::", "")
    n = n.split("(")[0][:46]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("q%-2d %8.1f -> %8.1f  (%6.1f us)  grid %7s  %s" % (qs.index(r["Queue_Id"]), s, e, e - s, r["Grid_Size_X"], n))
