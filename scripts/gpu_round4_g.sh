#!/bin/bash
# round 4, seventh GPU pass: factor-wave rotation A/B, fill variants A/B, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04g
mkdir -p $O
for rot in 1 0; do
  echo "ROT=$rot" >> $O/time.txt
  GPRF_POTRF_ROT=$rot timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
  GPRF_POTRF_ROT=$rot WORLD=8 TAG=shard8 timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
  GPRF_POTRF_ROT=$rot C4=1 timeout 600 python3 scripts/gpu_time.py 10 >> $O/time.txt 2>&1
done
for v in 0 1 2; do
  echo "FILL_VARIANT=$v" >> $O/time.txt
  GPRF_FILL_VARIANT=$v GPRF_FUSED_FILL=0 TAG=filled$v timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
done
grep -v amdgpu.ids $O/time.txt
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1
echo "all gpu tests rc=$?"; tail -6 $O/pytest_all.log
