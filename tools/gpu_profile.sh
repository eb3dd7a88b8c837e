#!/bin/bash
# rocprofv3 passes for bench.py: kernel trace + stats, then PMC passes (separately, as gpurun requires)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1 || { tail -5 $OUT/bench_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1 || { tail -5 $OUT/bench_pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1 || { tail -5 $OUT/bench_pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
