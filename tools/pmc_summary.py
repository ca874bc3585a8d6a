#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass: reads every *counter_collection.csv under DIR, prints
Kernel_Name,Dispatches,Counter_Name,Counter_Value_avg_KiB (the format of profiles/r0*/pmc_*.csv).  usage: pmc_summary.py DIR"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"], row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
print("Kernel_Name,Dispatches,Counter_Name,Counter_Value_avg_KiB")
for (name, ctr), (n, tot) in sorted(acc.items()):
    print(f"\"{name}\",{n},{ctr},{tot / n:.3f}")
