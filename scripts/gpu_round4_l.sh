#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04l
mkdir -p $O
timeout 600 python3 tests/diag/gpu_fill_compare.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 1200 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -25
