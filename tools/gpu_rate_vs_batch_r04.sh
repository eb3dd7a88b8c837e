#!/bin/bash
# blob_to_kzg_commitment and compute_blob_kzg_proof rates by batch size with one and two calls in flight (bench.py --in-flight):
# what the workspace slots buy callers whose batches do not fill the chip by themselves.  -> gpurun_out/r04/rate_vs_batch_in_flight.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
echo "[" > $O/rate_vs_batch_in_flight.json
first=1
for wl in commit proof; do
  for b in 512 1024 2048 4096 8192; do
    for f in 1 2; do
      line=$(python bench.py --workload $wl --batch $b --in-flight $f --steps 16 --warmup 4 --no-cpu-baseline --no-live-traffic --no-extra --blocking-setup 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'workload':'$wl','batch':$b,'in_flight':$f,'blobs_per_s':r['value'],'ms_per_step':r['ms_per_step']}))")
      [ $first -eq 1 ] || echo "," >> $O/rate_vs_batch_in_flight.json
      first=0
      echo "$line" >> $O/rate_vs_batch_in_flight.json
      echo "$line"
    done
  done
done
echo "]" >> $O/rate_vs_batch_in_flight.json
