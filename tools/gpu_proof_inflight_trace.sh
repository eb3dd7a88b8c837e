#!/bin/bash
# kernel timeline of compute_blob_kzg_proof at 4,096 blobs with F calls in flight (bench.py --in-flight F) under rocprofv3
# --kernel-trace: the last N dispatches into gpurun_out/proof_inflight<F>_timeline.txt   (usage: gpu_proof_inflight_trace.sh [F=2] [N=60])
set -o pipefail
R=$GRAFT_REPO_ROOT
F=${1:-2}
N=${2:-60}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p_trace -- python3 $R/bench.py --workload proof --in-flight $F --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --blocking-setup > $R/gpurun_out/p_trace_$F.log 2>&1
f=$(find $R/gpurun_out/p_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $f $N > $R/gpurun_out/proof_inflight${F}_timeline.txt
rm -rf $R/gpurun_out/p_trace
