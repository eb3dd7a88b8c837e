#!/bin/bash
# kernel timeline of single-blob commitments (device-resident) under rocprofv3 --kernel-trace: tools/trace_timeline.py of the
# last few dispatches into gpurun_out/single_commit_timeline.txt
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sc_trace -- python3 $R/tools/gpu_single_latency.py 0 20 commit > $R/gpurun_out/sc_trace.log 2>&1
f=$(find $R/gpurun_out/sc_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $f 12 > $R/gpurun_out/single_commit_timeline.txt
rm -rf $R/gpurun_out/sc_trace
