#!/bin/bash
# rocprofv3 passes of the north-star bench on the GPU box: kernel trace + three PMC passes (each in its own run, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes), condensed by scripts/prof_summary.py.  Usage (through gpurun):
#   bash scripts/profile_run.sh r02x
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --only-north-star --no-parity --steps 100 --warmup 10"
# (the trace pass shows the PRODUCT's launch structure: without counters rocprofv3 does not serialise the queues, and
# GPRF_DIAG=tool_env=0 keeps the library from switching to events / one queue because a tool is loaded; bounded all the same)
GPRF_DIAG=tool_env=0 GPRF_EVAL_TIMEOUT_S=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.log 2>&1
# counter passes: the profiler serialises the dispatches of ALL queues, and a stream-memory-operation wait in front of
# the solve (the join of the two Cholesky queues) then never sees its value written — the library puts the two Cholesky
# kernels on one queue when it sees ROCPROF_COUNTER_COLLECTION (set by --pmc); every pass under `timeout` all the same
export GPRF_DIAG=one_queue=1      # explicit: one queue for the counter passes (the library would also recognise the profiler)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/bench_pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/bench_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/bench_pmc_write.log 2>&1
python3 scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
git rev-parse --short HEAD > $OUT/commit.txt 2>/dev/null || true
# keep the merge-back small: the per-dispatch CSVs of the PMC passes are large
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/summary.txt | head -60
