#!/bin/bash
# rocprofv3 passes of the SE fill (k_fill_se: off the north-star path — K is generated inside the register Cholesky — so it is
# forced back with GPRF_DIAG fused_fill=0): kernel trace + SQ / FETCH_SIZE / WRITE_SIZE passes, each in its own run.
#   bash scripts/profile_fill.sh r05_fill
TAG=${1:-r05_fill}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export GPRF_DIAG=fused_fill=0
CMD="python3 scripts/gpu_time.py 40"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
export GPRF_DIAG=fused_fill=0,one_queue=1      # one queue for the counter passes
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python3 scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
python3 - <<PY >> $OUT/summary.txt
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join("$OUT", "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fill" in r.get("Kernel_Name", ""):
            acc["k_fill"][r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
print("== k_fill SQ counters, mean per launch ==")
for c, v in sorted(acc["k_fill"].items()):
    print("%-22s %.4g" % (c, v / max(n[c], 1)))
PY
python3 - <<PY
# counter bytes per launch / the kernel-trace pass's average duration of the SAME kernel -> fill_counters.json (bench.py)
import csv, glob, json, os
tr = json.load(open(os.path.join("$OUT", "traffic.json")))
avg_ns = None
for f in glob.glob(os.path.join("$OUT", "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fill" in r.get("Name", ""):
            avg_ns = float(r["AverageNs"])
k = tr.get("k_fill")
if k and avg_ns:
    d = {"kernel": "k_fill_se", "bytes_per_launch": k["bytes_per_launch"], "fetch_KiB": k["fetch_KiB"], "write_KiB": k["write_KiB"],
         "avg_ns": avg_ns, "counter_GBps": k["bytes_per_launch"] / avg_ns, "source_hash": tr.get("source_hash")}
    json.dump(d, open(os.path.join("$OUT", "fill_counters.json"), "w"), indent=1)
    print("fill counters:", d)
PY
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
grep "k_fill\|== k_fill" -A12 $OUT/summary.txt | head -60
