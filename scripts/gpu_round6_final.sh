#!/bin/bash
# round 6, the artefacts of a commit: the whole GPU suite, the numerics log, the rocprofv3 passes of the north-star bench, of
# the seismic shape and of the SE fill, the bench itself (with the counter traffic of THESE sources).
#   bash scripts/gpu_round6_final.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r06A}
O=gpurun_out/$TAG
mkdir -p $O
timeout 2700 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
echo "all gpu tests rc=$?"; tail -4 $O/pytest_gpu.log
timeout 900 python3 -m pytest tests/test_gpu_northstar.py tests/test_gpu_c4.py tests/test_gpu_trace_full.py -q -s -m gpu > $O/numerics.log 2>&1
echo "numerics rc=$?"; grep -aE "local_dist=|pair units vs|evaluations \(published" $O/numerics.log | grep -v print | cut -c1-400
bash scripts/profile_run.sh $TAG > $O/profile_run.log 2>&1
tail -45 $O/profile_run.log
cp gpurun_out/prof_$TAG/traffic.json profiles/r06_traffic.json 2>/dev/null
cp gpurun_out/prof_$TAG/summary.txt $O/rocprof_summary.txt 2>/dev/null
bash scripts/profile_c5.sh ${TAG}_c5 > $O/profile_c5.log 2>&1
cp gpurun_out/prof_${TAG}_c5/summary.txt $O/c5_rocprof_summary.txt 2>/dev/null
cp gpurun_out/prof_${TAG}_c5/traffic.json $O/c5_traffic.json 2>/dev/null
bash scripts/profile_fill.sh ${TAG}_fill > $O/profile_fill.log 2>&1
cp gpurun_out/prof_${TAG}_fill/summary.txt $O/fill_rocprof_summary.txt 2>/dev/null
cp gpurun_out/prof_${TAG}_fill/fill_counters.json profiles/r06_fill_counters.json 2>/dev/null
cp gpurun_out/prof_${TAG}_fill/fill_counters.json $O/ 2>/dev/null
timeout 1500 python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step_samples'], d['stages_ms'], d['roofline']['worst'], d['roofline']['traffic'], d.get('c4_evals_per_s'), d.get('c5_evals_per_s'), d['roofline'].get('fill_kernel'), d.get('big_units'), d.get('c5'), d.get('optimize_c3'))"
cp profiles/r06_traffic.json $O/ 2>/dev/null
# the round's authoritative record must carry the counter traffic of exactly these sources, an unextrapolated CPU baseline and
# the parity figures ("measurement hygiene"): a non-zero exit otherwise
python3 - "$O/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
bad = []
if r.get("traffic") is None or not r.get("traffic_sources_match"):
    bad.append("roofline.traffic is null or was profiled on other native sources: %s" % r.get("traffic_source"))
if "parity" not in d:
    bad.append("no parity object")
if "extrapolated" not in d.get("cpu_baseline", {}).get("sample", "") or "nothing extrapolated" not in d["cpu_baseline"]["sample"]:
    bad.append("cpu_baseline is not the whole evaluation")
print("FINAL CHECK:", "ok" if not bad else "; ".join(bad))
sys.exit(1 if bad else 0)
PY
