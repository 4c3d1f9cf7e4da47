cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05n
timeout 1200 python3 -m pytest tests/test_gpu_seismic.py tests/test_gpu_parity.py tests/test_gpu_neighbors.py -x -q -m gpu > gpurun_out/r05n/pytest_lld.log 2>&1; echo rc=$?; tail -4 gpurun_out/r05n/pytest_lld.log
(timeout 300 python3 scripts/gpu_seismic_time.py 20000 20; timeout 300 python3 scripts/gpu_seismic_time.py 100000 5) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05n/c5_time.txt
bash scripts/profile_big.sh r05_big 6000 | tail -30
