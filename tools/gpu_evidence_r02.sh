#!/bin/bash
# Round-2 evidence, everything except the rocprofv3 passes (tools/gpu_profile_r02.sh): bench lines, batch sweep,
# configs[4] per-GPU share, host-API rates, criterion-equivalent, two-rank gloo rehearsals.  Outputs: gpurun_out/r02/ev/
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/ev
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_default_c22.json 2> $O/bench_default.err || exit 1
python3 bench.py --workload proof --steps 10 --warmup 2 > $O/bench_proof4096_c22.json 2> $O/bench_proof.err || exit 1
python3 bench.py --workload verify --steps 10 --warmup 2 > $O/bench_verify65536_c22.json 2> $O/bench_verify.err || exit 1
for n in 1024 2048 4096 8192 16384; do
  python3 bench.py --batch $n --steps 6 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_batch$n.json 2> $O/bench_batch$n.err || exit 1
done
python3 bench.py --batch 131072 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/bench_commit_131072_per_gpu_c22.json 2> $O/bench_131072.err || exit 1
python3 bench.py --window-bits 16 --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_commit4096_c16_default_class.json 2> $O/bench_c16.err || exit 1
KATETH_AMD_COMB_GROUPS=4 python3 bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_commit4096_c22_g4.json 2> $O/bench_g4.err || exit 1
python3 tools/gpu_hostapi_bench.py 4096 22 > $O/hostapi_n4096_c22.json 2> $O/hostapi4096.err || exit 1
python3 tools/gpu_hostapi_bench.py 16384 22 > $O/hostapi_n16384_c22.json 2> $O/hostapi16384.err || exit 1
python3 tools/bench_criterion.py 16 > $O/criterion_equivalent_c16.json 2> $O/criterion.err || exit 1
python3 bench.py --gpus 2 --backend gloo --batch 512 --window-bits 16 --steps 2 --no-cpu-baseline > $O/bench_2ranks_gloo_one_card_commit.json 2> $O/gloo_commit.err || exit 1
python3 bench.py --gpus 2 --backend gloo --workload verify --batch 512 --window-bits 16 --steps 2 --no-cpu-baseline > $O/bench_2ranks_gloo_one_card_verify.json 2> $O/gloo_verify.err || exit 1
echo done
