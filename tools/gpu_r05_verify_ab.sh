#!/bin/bash
# Round 5: verify_blob_kzg_proof_batch at 65,536 triples, the library at HEAD against a previous build of it kept under tools/exp/ab/
# (git-ignored), same box, alternating.  -> gpurun_out/r05/vab_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="--workload verify --steps 12 --warmup 3 --no-cpu-baseline --no-live-traffic --blocking-setup --no-extra"
for rep in 1 2 3; do
  python bench.py $B > $O/vab_new_$rep.json 2>> $O/vab.err
  KATETH_AMD_LIB=$R/tools/exp/ab/libkateth_amd_prev.so python bench.py $B > $O/vab_old_$rep.json 2>> $O/vab.err
done
for f in $O/vab_*.json; do python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],3))" $f; done
echo ab done
