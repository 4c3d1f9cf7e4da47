#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box; outputs under gpurun_out/prof_<tag>/
# usage: scripts/prof_bench.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
python3 bench.py --steps 5 --warmup 2 --only-north-star "$@" > /dev/null 2>&1   # populate the input cache
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 50 --warmup 5 --only-north-star "$@" > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 10 --warmup 2 --only-north-star "$@" > $out/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 10 --warmup 2 --only-north-star "$@" > $out/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 bench.py --steps 10 --warmup 2 --only-north-star "$@" > $out/bench_pmc_write.log 2>&1
find $out -name "*.csv" | head -20
python3 scripts/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
