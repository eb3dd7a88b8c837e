#!/bin/bash
# rocprofv3 evidence for a round: usage  tools/gpu_profile.sh <tag>  (e.g. r05) -> gpurun_out/<tag>/prof/*; then
# `python tools/refresh_profiles.py <tag>` copies the summaries to be judged into profiles/<tag>/.  One script for every round
# (rounds 2-4 each had their own copy).  One process, --gpus 1, the table bench.py times (window_bits = 0 + KZG_CFG_TABLE_MAX).
# Kernel trace + stats for the default run and for each workload alone, then PMC passes -- each in its own run, NEVER combined with
# tracing -- for the three workloads (SQ counters, FETCH_SIZE, WRITE_SIZE).
# TRACE_ONLY=1: the traced runs only.   PMC_ONLY=1: the counter passes only.   IN_FLIGHT=1: also proofs with 2 and 3 calls in flight.
set -o pipefail
TAG=${1:?usage: gpu_profile.sh <tag>}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG/prof
mkdir -p $OUT
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  echo "[profile] $name" >&2
  rocprofv3 "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
}
B="--no-cpu-baseline --no-live-traffic --blocking-setup"
if [ -z "$PMC_ONLY" ]; then
run trace_default --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $R/bench.py --steps 10 --warmup 2 $B
run trace_commit --kernel-trace --stats --output-format csv -d $OUT/trace_commit -- python3 $R/bench.py --steps 10 --warmup 2 --no-extra $B
run trace_proof --kernel-trace --stats --output-format csv -d $OUT/trace_proof -- python3 $R/bench.py --workload proof --in-flight 1 --no-extra --steps 5 --warmup 3 $B
run trace_verify --kernel-trace --stats --output-format csv -d $OUT/trace_verify -- python3 $R/bench.py --workload verify --steps 5 --warmup 1 $B
TL="verify"
if [ -n "$IN_FLIGHT" ]; then
run trace_proof2 --kernel-trace --stats --output-format csv -d $OUT/trace_proof2 -- python3 $R/bench.py --workload proof --in-flight 2 --no-extra --steps 6 --warmup 3 $B
run trace_proof3 --kernel-trace --stats --output-format csv -d $OUT/trace_proof3 -- python3 $R/bench.py --workload proof --in-flight 3 --no-extra --steps 9 --warmup 3 $B
TL="verify proof2 proof3"
fi
for w in $TL proof; do
  f=$(find $OUT/trace_$w -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py $f 70 > $OUT/timeline_$w.txt
done
fi
[ -n "$TRACE_ONLY" ] && { python3 $R/tools/summarize_profiles.py $OUT; exit 0; }
SQ="SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
run pmc_verify_sq --pmc $SQ --output-format csv -d $OUT/pmc_verify_sq -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 $B
run pmc_verify_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_verify_fetch -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 $B
run pmc_verify_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_verify_write -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 $B
run pmc_commit_sq --pmc $SQ --output-format csv -d $OUT/pmc_commit_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra $B
run pmc_commit_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_commit_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra $B
run pmc_commit_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_commit_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-extra $B
run pmc_proof_sq --pmc $SQ --output-format csv -d $OUT/pmc_proof_sq -- python3 $R/bench.py --workload proof --in-flight 1 --no-extra --steps 3 --warmup 1 $B
python3 $R/tools/summarize_profiles.py $OUT
