#!/bin/bash
# Round 5: verify_blob_kzg_proof_batch at 65,536 triples with the lincombs' scalars GLV-split (default) and unsplit
# (KATETH_AMD_VAR_GLV=0), same box, alternating.  -> gpurun_out/r05/glv_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "variable_base or verify or group_dev or workspace" > $O/glv_tests.log 2>&1
echo "pytest rc=$?" >> $O/glv_tests.log
tail -3 $O/glv_tests.log
B="--workload verify --steps 20 --warmup 4 --no-cpu-baseline --no-live-traffic --window-bits 16"
for rep in 1 2; do
  python bench.py $B > $O/glv_on_$rep.json 2>> $O/glv.err
  KATETH_AMD_VAR_GLV=0 python bench.py $B > $O/glv_off_$rep.json 2>> $O/glv.err
done
python bench.py --workload proof --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic > $O/proof_after_slot_fix.json 2>> $O/glv.err
echo glv done
