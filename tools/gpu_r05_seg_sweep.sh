#!/bin/bash
# Round 5: verify at 65,536 triples over the balanced bucket kernel's share size (KATETH_AMD_VAR_SEG=<entries per lane>; 1 = the engine's
# choice, 33 on an MI355X).  -> gpurun_out/r05/segsweep.txt
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="--workload verify --steps 12 --warmup 3 --no-cpu-baseline --no-live-traffic --blocking-setup --no-extra"
: > $O/segsweep.txt
for rep in 1 2; do
for E in 1 26 29 31 33 36 40 48; do
  KATETH_AMD_VAR_SEG=$E python bench.py $B 2>> $O/seg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('E=$E rep=$rep', round(d['value']), round(d['ms_per_step'],3))" | tee -a $O/segsweep.txt
done
done
