#!/bin/bash
# kernel timeline of one 65,536-triple verify_blob_kzg_proof_batch call under rocprofv3 --kernel-trace: tools/trace_timeline.py
# of the last call's dispatches into gpurun_out/verify_timeline.txt   (usage: gpu_verify_trace.sh [dispatches=40])
set -o pipefail
R=$GRAFT_REPO_ROOT
N=${1:-40}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/v_trace -- python3 $R/bench.py --workload verify --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/v_trace.log 2>&1
f=$(find $R/gpurun_out/v_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $f $N > $R/gpurun_out/verify_timeline.txt
rm -rf $R/gpurun_out/v_trace
