#!/bin/bash
# round 4, third GPU pass: run-ahead Cholesky with wave 0's tiles front-loaded; strip-wise k_fill
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
for ra in 0 1; do
  GPRF_POTRF_RA=$ra timeout 300 python3 scripts/gpu_time.py 40 > $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra WORLD=8 TAG=shard8 timeout 300 python3 scripts/gpu_time.py 40 >> $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra C4=1 timeout 600 python3 scripts/gpu_time.py 10 >> $O/time_ra$ra.txt 2>&1
  GPRF_POTRF_RA=$ra GPRF_POTRF_DUAL=2 TAG=oneq timeout 300 python3 scripts/gpu_time.py 40 >> $O/time_ra$ra.txt 2>&1
  grep -v amdgpu.ids $O/time_ra$ra.txt
done
GPRF_FUSED_FILL=0 TAG=filled timeout 300 python3 scripts/gpu_time.py 40 2>&1 | grep -v amdgpu.ids | tee $O/time_fill.txt
# in-kernel stamps of the run-ahead form (diagnostic build)
for st in 1 2; do
  for ra in 1 0; do
    echo "== GPRF_POTRF_STAMPS=$st RA=$ra" >> $O/stamps.txt
    GPRF_LIB=/root/repo/build_variants/libgprf_profile.so GPRF_BUILD_DEFS=-DGPRF_PROFILE GPRF_POTRF_RA=$ra GPRF_POTRF_STAMPS=$st timeout 600 python3 scripts/gpu_potrf_stamps.py 2>&1 | grep -v amdgpu.ids >> $O/stamps.txt
  done
done
cat $O/stamps.txt
