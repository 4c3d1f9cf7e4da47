#!/bin/bash
# round 4, fifth GPU pass: units of more than 1024 points; workgroup trace of the Cholesky; accuracy of the old substitution panel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04e
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_big_units.py -x -q -m gpu -s > $O/pytest_big.log 2>&1
echo "big units rc=$?"; grep -v amdgpu.ids $O/pytest_big.log | tail -15
timeout 1500 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_multidev.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest_rest.log 2>&1
echo "rest rc=$?"; tail -5 $O/pytest_rest.log
GPRF_LIB=/root/repo/build_variants/libgprf_wgtrace5.so GPRF_BUILD_DEFS=-DGPRF_WGTRACE=5 KERNEL=5 GPRF_POTRF_RA=0 timeout 600 python3 scripts/gpu_wg_trace.py 2>&1 | grep -v amdgpu.ids > $O/wgtrace5_ra0.txt
head -40 $O/wgtrace5_ra0.txt
GPRF_LIB=/root/repo/build_variants/libgprf_subst.so GPRF_POTRF_RA=0 timeout 1500 python3 -m pytest tests/test_gpu_northstar.py -q -m gpu -s -k "pair_units or gradient_against" > $O/northstar_subst.log 2>&1
echo "northstar(subst) rc=$?"; grep "pair units vs\|local_dist=\|passed\|failed" $O/northstar_subst.log
