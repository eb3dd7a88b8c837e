#!/bin/bash
# verify / commit rate against the batch size at HEAD (device-resident, one box): one JSON object per line into
# gpurun_out/rate_vs_batch.jsonl   (bench.py --workload W --batch N --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic)
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rate_vs_batch.jsonl
: > $OUT
for n in 1024 4096 8192 16384 32768 49152 65536; do
  timeout -k 10 200 python3 $R/bench.py --workload verify --batch $n --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic >> $OUT 2>/dev/null || exit 1
done
for n in 1024 2048 8192 16384 131072; do
  timeout -k 10 300 python3 $R/bench.py --workload commit --batch $n --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-extra >> $OUT 2>/dev/null || exit 1
done
timeout -k 10 300 python3 $R/bench.py --workload proof --batch 16384 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic >> $OUT 2>/dev/null || exit 1
