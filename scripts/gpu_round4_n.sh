#!/bin/bash
# round 4, pass N: the eight-wave Cholesky for units of up to 20 tiles — variants, seismic tests, C5 stage times (A/B)
mkdir -p gpurun_out/r04n
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_seismic.py tests/test_gpu_parity.py tests/test_gpu_big_units.py -x -q -m gpu > gpurun_out/r04n/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r04n/tests.log
timeout 300 python scripts/gpu_seismic_time.py 20000 20 > gpurun_out/r04n/c5_new.log 2>&1
GPRF_POTRF_BIG8=0 timeout 300 python scripts/gpu_seismic_time.py 20000 20 > gpurun_out/r04n/c5_old.log 2>&1
GPRF_POTRF_REG=0 timeout 300 python scripts/gpu_seismic_time.py 20000 20 > gpurun_out/r04n/c5_generic.log 2>&1
tail -5 gpurun_out/r04n/tests.log; grep "stages\|sync" gpurun_out/r04n/c5_*.log
