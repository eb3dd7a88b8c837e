#!/bin/bash
# Round-3 evidence (same passes as round 2, library default table class = automatic): rocprofv3 passes over bench.py (one process, --gpus 1).  Kernel trace + stats for the default run
# (commit + proof + verify at c = 16), then PMC passes -- each in its own run, never combined with tracing -- for the
# verify workload (SQ counters, FETCH_SIZE, WRITE_SIZE) and for the commit workload.  Summaries land in
# gpurun_out/r03/prof/*.json|csv; copy the ones to be judged into profiles/r03/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/prof
mkdir -p $OUT
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  echo "[profile] $name" >&2
  rocprofv3 "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
}
WIN=${WIN:-0}
run trace_default --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --window-bits $WIN
run trace_commit --kernel-trace --stats --output-format csv -d $OUT/trace_commit -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-live-traffic --window-bits $WIN
run trace_proof --kernel-trace --stats --output-format csv -d $OUT/trace_proof -- python3 $R/bench.py --workload proof --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --window-bits $WIN
run trace_verify --kernel-trace --stats --output-format csv -d $OUT/trace_verify -- python3 $R/bench.py --workload verify --steps 5 --warmup 1 --no-cpu-baseline --no-live-traffic --window-bits $WIN
[ -n "$TRACE_ONLY" ] && { python3 $R/tools/summarize_profiles.py $OUT; exit 0; }
run pmc_verify_sq --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_verify_sq -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --window-bits $WIN
run pmc_verify_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_verify_fetch -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --window-bits $WIN
run pmc_verify_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_verify_write -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --window-bits $WIN
run pmc_commit_sq --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_commit_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-live-traffic --window-bits $WIN
run pmc_commit_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_commit_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-live-traffic --window-bits $WIN
run pmc_commit_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_commit_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-live-traffic --window-bits $WIN
run pmc_proof_sq --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_proof_sq -- python3 $R/bench.py --workload proof --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --window-bits $WIN
python3 $R/tools/summarize_profiles.py $OUT
