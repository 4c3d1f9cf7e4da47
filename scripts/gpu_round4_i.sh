#!/bin/bash
# round 4, ninth GPU pass: k_fill_se; parity of the filled path; timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04i
mkdir -p $O
for v in 0 4; do
  echo "FILL_VARIANT=$v" >> $O/time.txt
  GPRF_FILL_VARIANT=$v GPRF_FUSED_FILL=0 TAG=filled$v timeout 300 python3 scripts/gpu_time.py 40 >> $O/time.txt 2>&1
done
grep -v amdgpu.ids $O/time.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_big_units.py tests/test_gpu_seismic.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
GPRF_FUSED_FILL=0 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_filled.log 2>&1
echo "pytest (K pool forced) rc=$?"; tail -4 $O/pytest_filled.log
