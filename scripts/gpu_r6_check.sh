#!/bin/bash
# round 6, a quick pass on the GPU box: the whole GPU suite, then the bench (default flags).   bash scripts/gpu_r6_check.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r06a}
O=gpurun_out/$TAG
mkdir -p $O
timeout 2700 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
echo "all gpu tests rc=$?"; tail -6 $O/pytest_gpu.log
timeout 1500 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -3 $O/bench.err; python3 -c "
import json;d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step_samples'], d['stages_ms'], d['roofline']['worst'], d['roofline']['traffic'], d['roofline'].get('traffic_sources_match'), d.get('parity'), d.get('cpu_baseline'), d.get('c4_evals_per_s'), d.get('c5_evals_per_s'))"
