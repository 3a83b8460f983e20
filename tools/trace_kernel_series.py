#!/usr/bin/env python3
"""Print the duration series of one kernel from a rocprofv3 --kernel-trace CSV (usage: trace_kernel_series.py <dir> <kernel-substring>)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
d = [x[1] / 1e3 for x in rows]
print("n", len(d), "mean %.1f us  min %.1f  max %.1f" % (sum(d) / len(d), min(d), max(d)))
for i in range(0, len(d), 33):
    blk = d[i:i + 33]
    print("launches %4d-%4d: mean %.1f  min %.1f  max %.1f" % (i, i + len(blk) - 1, sum(blk) / len(blk), min(blk), max(blk)))
