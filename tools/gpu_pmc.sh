#!/bin/bash
# rocprofv3 PMC passes over tools/gpu_probe.py (one pass per counter group; never combined with tracing)
# usage: [PROG=tools/gpu_stage_bench.py KERNELS="k_challenge|k_eval"] tools/gpu_pmc.sh "<program args>" "CNT_A CNT_B" "CNT_C" ...
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
ARGS=$1
shift
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/${PROG:-tools/gpu_probe.py} $ARGS > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:40], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        import re
        if re.search(r"${KERNELS:-msm_fixed}", kn):
            print("%-42s %-28s per-launch %.4g (launches %d)" % (kn, cn, v / n, n))
PY
