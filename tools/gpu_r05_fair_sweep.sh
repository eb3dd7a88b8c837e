#!/bin/bash
# Round 5: blob_to_kzg_commitment over the period with which the two waves of a SIMD trade issue priority (KATETH_AMD_COMB_FAIR = log2 of
# shader cycles; round 2 chose 20 for the radix-2^28 kernel) -- re-checked on the shorter additions of the radix-2^30 kernel.
# -> gpurun_out/r05/fairsweep.txt
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="--no-extra --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --blocking-setup"
: > $O/fairsweep.txt
for rep in 1 2; do
for F in 20 17 18 19 21 22 23; do
  KATETH_AMD_COMB_FAIR=$F python bench.py $B 2>> $O/fair.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fair=$F rep=$rep', round(d['value']), round(d['ms_per_step'],3))" | tee -a $O/fairsweep.txt
done
done
