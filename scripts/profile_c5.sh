#!/bin/bash
# rocprofv3 passes of BASELINE configs[4]'s shape (seismic stand-in, n = 20000: lld / Matern-3/2, split-tree blocks < 210,
# threshold 0.6, both gradients): kernel trace + FETCH_SIZE / WRITE_SIZE passes (each in its own run), condensed by
# scripts/prof_summary.py.  Usage (through gpurun):   bash scripts/profile_c5.sh r03_c5
TAG=${1:-r03_c5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 scripts/gpu_seismic_time.py 20000 20"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
export GPRF_DIAG=one_queue=1      # one queue for the counter passes
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python3 scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
cat $OUT/trace.log | tail -5
cat $OUT/summary.txt | head -70
