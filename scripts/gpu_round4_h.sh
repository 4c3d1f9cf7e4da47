#!/bin/bash
# round 4, eighth GPU pass: fill with non-temporal stores; fill PMC passes; bench with the big-units leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
for v in 0 3; do
  echo "FILL_VARIANT=$v" >> $O/time.txt
  GPRF_FILL_VARIANT=$v GPRF_FUSED_FILL=0 TAG=filled$v timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
done
grep -v amdgpu.ids $O/time.txt
bash scripts/profile_fill.sh r04h_fill 0 > $O/profile_fill.log 2>&1
tail -40 $O/profile_fill.log
timeout 1500 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench.json'))
print(d['value'], d['stages_ms'], d['roofline']['worst'], d.get('c4_evals_per_s'), d.get('c5_evals_per_s'), d.get('big_units'))"
tail -3 $O/bench.err
