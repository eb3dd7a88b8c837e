#!/bin/bash
# Round 5: blob_to_kzg_commitment with the fp30 MSM kernel (HEAD) against the fp28 one (the previous commit's library,
# tools/exp/ab/libkateth_amd_fp28.so), same box, alternating, plus the microbenchmark.  -> gpurun_out/r05/ab_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="--no-extra --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --blocking-setup"
for rep in 1 2; do
  python bench.py $B > $O/ab_commit_fp30_$rep.json 2>> $O/ab.err
  KATETH_AMD_LIB=$R/tools/exp/ab/libkateth_amd_fp28.so python bench.py $B > $O/ab_commit_fp28_$rep.json 2>> $O/ab.err
done
python bench.py $B --batch 16384 --steps 6 --warmup 2 > $O/ab_commit16384_fp30.json 2>> $O/ab.err
KATETH_AMD_LIB=$R/tools/exp/ab/libkateth_amd_fp28.so python bench.py $B --batch 16384 --steps 6 --warmup 2 > $O/ab_commit16384_fp28.json 2>> $O/ab.err
tools/exp/fp30_bench $O/fp30_bench_same_box.json > $O/fp30_bench_same_box.txt 2>&1
echo ab done
