#!/bin/bash
# round 4, first GPU pass: the plumbing (self-launching bench, group fallback, io modes) + baseline numbers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_multidev.py tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_optimize.py -x -q -m gpu > $O/pytest_a.log 2>&1
echo "pytest rc=$?" >> $O/pytest_a.log
tail -5 $O/pytest_a.log
timeout 900 python3 bench.py --steps 200 --warmup 20 --no-c4 --no-c5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
timeout 1500 python3 scripts/gpu_io_mode_ab.py 3 > $O/io_mode_ab.txt 2>&1
cat $O/io_mode_ab.txt
GPRF_BUILD_DEFS=-DGPRF_PROFILE true
