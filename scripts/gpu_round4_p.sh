#!/bin/bash
# round 4, pass P: completion word folded into the assembly — variants + headline A/B
mkdir -p gpurun_out/r04p
timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_multidev.py -x -q -m gpu > gpurun_out/r04p/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r04p/tests.log
for rep in 1 2; do
timeout 600 python bench.py --only-north-star > gpurun_out/r04p/bench_fold_$rep.json 2> gpurun_out/r04p/bench_fold_$rep.err
GPRF_DONE_FOLD=0 timeout 600 python bench.py --only-north-star > gpurun_out/r04p/bench_nofold_$rep.json 2> gpurun_out/r04p/bench_nofold_$rep.err
done
tail -3 gpurun_out/r04p/tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04p/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['ms_per_step_samples'])
    except Exception as e: print(f, 'ERR', e)
PY
