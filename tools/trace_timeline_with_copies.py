#!/usr/bin/env python3
"""Timeline of the LAST call in a rocprofv3 --kernel-trace --memory-copy-trace output directory: kernels and memory copies merged
by start time (copies of at least `min_kib` KiB), relative to the first row shown.
usage: trace_timeline_with_copies.py <dir> [count=60] [min_kib=64]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
min_kib = float(sys.argv[3]) if len(sys.argv) > 3 else 64
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%-3s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:72])))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        size = float(r.get("Size", r.get("Bytes", 0)) or 0)
        if size >= min_kib * 1024:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %.1f MiB" % (r.get("Direction", r.get("Kind", "?")), size / 2**20)))
rows.sort()
rows = rows[-count:]
t0 = rows[0][0]
for a, b, what in rows:
    print("%9.3f ms  +%8.3f ms  %s" % ((a - t0) / 1e6, (b - a) / 1e6, what))
