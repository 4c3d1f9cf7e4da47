#!/bin/bash
# round 6, a quick look at a kernel change: the bit-for-bit variants, the per-stage parity tests, a short bench with stage times.
#   bash scripts/gpu_r6_quick.sh <tag> [pytest -k expression]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-q}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -q -m gpu -x ${2:+-k "$2"} > $O/pytest_quick.log 2>&1
echo "quick tests rc=$?"; tail -12 $O/pytest_quick.log
for d in ${DIAGS:-""}; do
GPRF_DIAG=$d timeout 600 python3 bench.py --only-north-star --no-parity --steps 200 --warmup 20 > $O/bench_$d.json 2> $O/bench_$d.err
echo "bench [$d] rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench_$d.json'))
print(d['value'], d['ms_per_step_samples'], d['stages_ms'])"
done
