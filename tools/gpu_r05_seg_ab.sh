#!/bin/bash
# Round 5: verify_blob_kzg_proof_batch at 65,536 triples with the balanced bucket kernel (default: k_var_buckets_seg, equal shares of
# the sorted entry list per lane) against one thread per bucket (KATETH_AMD_VAR_SEG=0: k_var_buckets_flat), same box, alternating.
# -> gpurun_out/r05/seg_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="--workload verify --steps 12 --warmup 3 --no-cpu-baseline --no-live-traffic --blocking-setup --no-extra"
for rep in 1 2; do
  python bench.py $B > $O/seg_verify_seg_$rep.json 2>> $O/seg.err
  KATETH_AMD_VAR_SEG=0 python bench.py $B > $O/seg_verify_perbucket_$rep.json 2>> $O/seg.err
done
for f in $O/seg_verify_*.json; do python -c "
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],3))" $f; done
echo ab done
