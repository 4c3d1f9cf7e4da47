#!/bin/bash
# round 4, tenth GPU pass: grouped part-major walk of the solve / gradient grids: stage times and HBM traffic per group size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04j
mkdir -p $O
for g in 0 32 64 128; do
  echo "PM_GROUP=$g" >> $O/time.txt
  GPRF_PM_GROUP=$g timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
  GPRF_PM_GROUP=$g GPRF_PART_MAJOR=1 C4=1 TAG=C4pm timeout 600 python3 scripts/gpu_time.py 10 >> $O/time.txt 2>&1
done
GPRF_PART_MAJOR=0 TAG=unitmajor timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
grep -v amdgpu.ids $O/time.txt
export GPRF_POTRF_DUAL=2
for g in 0 64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GPRF_PM_GROUP=$g timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${g}_$c -- python3 scripts/gpu_time.py 20 > $O/pmc_${g}_$c.log 2>&1
  done
  python3 - <<PY
import csv, glob, os
from collections import defaultdict
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(float); n = defaultdict(int)
    for f in glob.glob(os.path.join("$O", "pmc_${g}_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for nm in ("k_mgrad", "k_solve_panel", "k_at_wide", "k_potrf_reg2", "k_potrf_reg8"):
                if nm in k:
                    acc[nm] += float(r["Counter_Value"]); n[nm] += 1
    print("PM_GROUP=$g", c, {k: round(v / n[k] / 1024.0, 1) for k, v in acc.items()}, "MiB per launch (FETCH: x2 for bytes)")
PY
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
