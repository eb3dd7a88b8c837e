#!/bin/bash
# usage: tools/gpu_evidence.sh <tag>   (e.g. r05).  Evidence besides tools/gpu_profile.sh: host-buffer rates, the library-default
# budget, proofs with one, two and three calls in flight, the criterion-equivalent single-item table, the host-buffer proof call's
# timeline (kernels + PCIe copies), the group path.  Outputs under gpurun_out/<tag>/; copied into profiles/<tag>/ by hand.
set -o pipefail
TAG=${1:?usage: gpu_evidence.sh <tag>}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/gpu_hostapi_bench.py 4096 0 > $O/hostapi_n4096.json 2> $O/evidence.err
python tools/gpu_hostapi_bench.py 16384 0 > $O/hostapi_n16384.json 2>> $O/evidence.err
python bench.py --default-budget --no-extra --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic > $O/bench_commit_default_budget.json 2>> $O/evidence.err
python bench.py --workload proof --in-flight 2 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/bench_proof_two_in_flight.json 2>> $O/evidence.err
python bench.py --workload proof --in-flight 3 --steps 21 --warmup 6 --no-cpu-baseline --no-live-traffic > $O/bench_proof_three_in_flight.json 2>> $O/evidence.err
python bench.py --workload proof --in-flight 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > $O/bench_proof_one_in_flight.json 2>> $O/evidence.err
python tools/bench_criterion.py 0 --no-cpu > $O/criterion_equivalent.json 2>> $O/evidence.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/hp_trace -- python3 $R/tools/gpu_host_proof_trace.py 4096 3 > $O/host_proof_trace.log 2>&1
python3 $R/tools/trace_timeline_with_copies.py $O/hp_trace 40 > $O/host_proof_4096_timeline.txt
rm -rf $O/hp_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_proof -- python3 $R/bench.py --workload proof --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --blocking-setup > $O/trace_proof.log 2>&1
cp $(find $O/trace_proof -name "*kernel_stats.csv" | head -1) $O/trace_proof4096_kernel_stats.csv
rm -rf $O/trace_proof
cd $R
python bench.py --group 2 --batch 2048 --steps 10 --warmup 3 > $O/group2_commit.json 2>> $O/evidence.err
python bench.py --group 2 --batch 2048 --workload proof --steps 10 --warmup 3 > $O/group2_proof.json 2>> $O/evidence.err
python bench.py --group 2 --batch 32768 --workload verify --steps 5 --warmup 2 > $O/group2_verify.json 2>> $O/evidence.err
echo evidence done
