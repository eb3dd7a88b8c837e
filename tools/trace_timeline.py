#!/usr/bin/env python3
"""Timeline of the LAST `count` kernel dispatches of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the
previous end, queue.  usage: trace_timeline.py <kernel_trace.csv> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f ms  +%8.3f ms  gap %8.3f  q%-3s %s" % ((a - t0) / 1e6, (b - a) / 1e6, (a - prev_end) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    prev_end = max(prev_end, b)
