#!/bin/bash
# round 4, sixth GPU pass: big units (fixed), trimmed K generation, RA off by default; timings; full suite + bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04f
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_big_units.py -x -q -m gpu -s > $O/pytest_big.log 2>&1
echo "big units rc=$?"; grep -v amdgpu.ids $O/pytest_big.log | tail -12
timeout 300 python3 scripts/gpu_time.py 40 > $O/time.txt 2>&1
WORLD=8 TAG=shard8 timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
C4=1 timeout 600 python3 scripts/gpu_time.py 10 >> $O/time.txt 2>&1
GPRF_POTRF_DUAL=2 TAG=oneq timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
grep -v amdgpu.ids $O/time.txt
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1
echo "all gpu tests rc=$?"; tail -6 $O/pytest_all.log
timeout 1200 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step_samples'], d['stages_ms'], d['roofline']['worst'], d.get('c4_evals_per_s'), d.get('c5_evals_per_s'), d['roofline'].get('fill_kernel'))"
