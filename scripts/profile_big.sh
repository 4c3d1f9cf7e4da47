#!/bin/bash
# rocprofv3 passes of the blocked path (units beyond one workgroup): ONE block of n points (default 6000: the per-dispatch CSVs of
# the counter pass stay small), kernel trace + the SQ pass (MFMA-busy of k_big_gemm).   bash scripts/profile_big.sh r05_big [n]
TAG=${1:-r05_big}
N=${2:-6000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 scripts/gpu_big_units_time.py $N 1 0 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
export GPRF_DIAG=one_queue=1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
print("== rocprofv3 --kernel-trace --stats -- $CMD ==")
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["TotalDurationNs"]) > 2e5:
            print("%-64s calls %6s avg_us %10.1f total_ms %9.2f pct %s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].split("(")[0][-44:]
        agg[nm][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[nm][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-44:]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("== SQ pass (dispatches serialised): per-launch averages; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz) ==")
for nm in sorted(agg):
    if "big" in nm or "mgrad" in nm or "k_at" in nm:
        d = sum(dur[nm]) / max(len(dur[nm]), 1)
        mb = agg[nm]["SQ_VALU_MFMA_BUSY_CYCLES"] / max(cnt[nm]["SQ_VALU_MFMA_BUSY_CYCLES"], 1)
        print("%-44s launches %5d avg_us %9.1f  MFMA-busy %5.1f %%  " % (nm, len(dur[nm]), d / 1e3, 100 * mb / (1024 * d * 2.4)) +
              "  ".join("%s=%.4g" % (c, agg[nm][c] / cnt[nm][c]) for c in sorted(agg[nm])))
PY
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.txt
