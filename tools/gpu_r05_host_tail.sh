#!/bin/bash
# Round 5, the host tail of a verification call: verify_blob_kzg_proof_batch at 65,536 triples and the single-item table with the
# host's Fp product on mulx / adcx / adox (default) and on the portable loop (KATETH_AMD_HOST_FP=portable), same box, same process
# shape, alternating.  -> gpurun_out/r05/host_tail_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for rep in 1 2; do
  python bench.py --workload verify --steps 20 --warmup 4 --no-cpu-baseline --no-live-traffic --window-bits 16 > $O/host_tail_verify_mulx_$rep.json 2>> $O/host_tail.err
  KATETH_AMD_HOST_FP=portable python bench.py --workload verify --steps 20 --warmup 4 --no-cpu-baseline --no-live-traffic --window-bits 16 > $O/host_tail_verify_portable_$rep.json 2>> $O/host_tail.err
done
python tools/bench_criterion.py 16 --no-cpu > $O/host_tail_criterion_mulx.json 2>> $O/host_tail.err
KATETH_AMD_HOST_FP=portable python tools/bench_criterion.py 16 --no-cpu > $O/host_tail_criterion_portable.json 2>> $O/host_tail.err
grep -m1 "model name" /proc/cpuinfo > $O/host_cpu.txt; nproc >> $O/host_cpu.txt
echo host tail done
