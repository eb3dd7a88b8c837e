#!/bin/bash
# VERDICT r02 "next" 6: verify_blob_kzg_proof_batch at 65,536 with point decoding beside the full-chip hash (both kernels
# NOTE: needs the experiment commit named in profiles/r03/verify_cohash_traded_priority_rejected.json (the kernel and the knob were removed from the product afterwards).
# trading issue priority), against the default (hash alone, then decoding || evaluation), same box, alternating runs.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/cohash
mkdir -p $OUT
for rep in 1 2; do
  python3 $R/bench.py --workload verify --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > $OUT/default_$rep.json 2>> $OUT/err.log || exit 1
  for s in 20 18 22; do
    KATETH_AMD_VERIFY_COHASH=$s python3 $R/bench.py --workload verify --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > $OUT/cohash${s}_$rep.json 2>> $OUT/err.log || exit 1
  done
done
# (the parity test ran with the knob set at the experiment commit)
python3 - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_by_class_per_call"]
    out[os.path.basename(f)[:-5]] = {"blobs_per_s": round(d["value"]), "ms_per_call": round(d["ms_per_step"], 3), "kernel_ms": {a: round(b, 3) for a, b in k.items()}}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
