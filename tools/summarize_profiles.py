#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/gpu_profile.sh: the kernel-stats CSV of the traced run and, per PMC pass,
per-kernel per-launch counter averages (launch counts included) -> <dir>/summary_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("kzg::", "").replace("void ", "")
    return name.strip()[:60]


def main(out):
    for name in ("trace_default", "trace_commit", "trace_proof", "trace_proof2", "trace_proof3", "trace_verify"):
        for f in glob.glob(os.path.join(out, name, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(out, "summary_%s_kernel_stats.csv" % name))
    summary = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
        summary[os.path.basename(d)] = {k: {c: {"per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()} for k, cs in acc.items()
                                        if re.search(r"msm_fixed|comb|challenge|eval_frac|decompress|var_|poly|reduce|compress|transcript|batch_", k)}
    json.dump(summary, open(os.path.join(out, "summary_pmc.json"), "w"), indent=1, sort_keys=True)
    for name, kernels in summary.items():
        for k, cs in sorted(kernels.items()):
            print("%-18s %-44s %s" % (name, k, "  ".join("%s=%.4g(x%d)" % (c, v["per_launch"], v["launches"]) for c, v in sorted(cs.items()))))


if __name__ == "__main__":
    main(sys.argv[1])
