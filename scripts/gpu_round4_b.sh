#!/bin/bash
# round 4, second GPU pass: the run-ahead Cholesky — bit-compare against the barrier form, then timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu > $O/pytest_variants.log 2>&1
echo "variants rc=$?"; tail -4 $O/pytest_variants.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_optimize.py -x -q -m gpu > $O/pytest_parity.log 2>&1
echo "parity rc=$?"; tail -4 $O/pytest_parity.log
for ra in 0 1; do
  GPRF_POTRF_RA=$ra timeout 300 python3 scripts/gpu_time.py 40 > $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra WORLD=8 TAG=shard8 timeout 300 python3 scripts/gpu_time.py 40 >> $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra C4=1 timeout 600 python3 scripts/gpu_time.py 10 >> $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra GPRF_POTRF_DUAL=2 TAG=oneq timeout 300 python3 scripts/gpu_time.py 40 >> $O/time_ra$ra.txt 2>&1
  cat $O/time_ra$ra.txt
  GPRF_POTRF_RA=$ra timeout 600 python3 bench.py --only-north-star --steps 200 --warmup 20 > $O/bench_ra$ra.json 2> $O/bench_ra$ra.err
  python3 -c "import json;d=json.load(open('$O/bench_ra$ra.json'));print('RA=$ra', d['value'], d['ms_per_step_samples'], d['stages_ms'])"
done
