#!/bin/bash
# Round 5: blob_to_kzg_commitment with the library at HEAD against a previous build of it kept under tools/exp/ab/ (git-ignored; the
# adder with carry passes, then the one with injected differences before ZZ / ZZZ went U-form), same box, alternating.
# -> gpurun_out/r05/inj_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="--no-extra --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --blocking-setup"
for rep in 1 2; do
  python bench.py $B > $O/inj_commit_new_$rep.json 2>> $O/inj.err
  KATETH_AMD_LIB=$R/tools/exp/ab/libkateth_amd_inj.so python bench.py $B > $O/inj_commit_old_$rep.json 2>> $O/inj.err
done
echo ab done
