#!/bin/bash
# Round 5 soak at HEAD (the fixed-base MSM on the new signed radix-2^30 field is what it is for): the comb kernel against round 1's
# window-table kernel on 12 x 32-bit limbs (independent field code), launch shapes + host pipelines against the class-8 engine,
# commit -> prove -> verify across the batch-size regimes, calls in flight.  -> gpurun_out/r05/soak_*.json|log
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout -k 10 400 python tools/gpu_soak_msm.py 60 22 > $O/soak_msm_c22.log 2>&1; echo "rc=$?" >> $O/soak_msm_c22.log; tail -n 3 $O/soak_msm_c22.log
timeout -k 10 200 python tools/gpu_soak_msm.py 30 16 > $O/soak_msm_c16.log 2>&1; echo "rc=$?" >> $O/soak_msm_c16.log; tail -n 2 $O/soak_msm_c16.log
timeout -k 10 500 python tools/gpu_soak_shapes.py 8 > $O/soak_shapes.json 2> $O/soak_shapes.err; echo "shapes rc=$?"; tail -c 400 $O/soak_shapes.json
timeout -k 10 300 python tools/gpu_soak_verify.py 3 22 > $O/soak_verify.log 2>&1; echo "verify rc=$?"; tail -n 2 $O/soak_verify.log
timeout -k 10 120 python tools/gpu_soak_inflight.py 120 > $O/soak_inflight.json 2> $O/soak_inflight.err; echo "inflight rc=$?"; tail -c 300 $O/soak_inflight.json
echo soak done
