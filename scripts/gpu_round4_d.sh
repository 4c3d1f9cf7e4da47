#!/bin/bash
# round 4, fourth GPU pass: row panel on the matrix pipe (V_jj in the loop) — parity, accuracy against 80 bits, timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
for ra in 1 0; do
  GPRF_POTRF_RA=$ra timeout 300 python3 scripts/gpu_time.py 40 > $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra WORLD=8 TAG=shard8 timeout 300 python3 scripts/gpu_time.py 40 >> $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra C4=1 timeout 600 python3 scripts/gpu_time.py 10 >> $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra GPRF_POTRF_DUAL=2 TAG=oneq timeout 300 python3 scripts/gpu_time.py 40 >> $O/time_ra$ra.txt 2>&1
  echo "RA=$ra"; grep -v amdgpu.ids $O/time_ra$ra.txt
done
timeout 1500 python3 -m pytest tests/test_gpu_northstar.py -x -q -m gpu -s > $O/northstar.log 2>&1
echo "northstar rc=$?"; grep -v amdgpu.ids $O/northstar.log | tail -12
