#!/bin/bash
# round 4, pass O: lld / Matern kernel math (polynomial great-circle terms), k_solve_panel<20,2>
mkdir -p gpurun_out/r04o
timeout 1500 python -m pytest tests/test_gpu_seismic.py tests/test_gpu_variants.py tests/test_gpu_neighbors.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04o/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r04o/tests.log
timeout 300 python scripts/gpu_seismic_time.py 20000 20 > gpurun_out/r04o/c5_new.log 2>&1
tail -3 gpurun_out/r04o/tests.log; grep "stages\|sync\|callback" gpurun_out/r04o/c5_*.log
