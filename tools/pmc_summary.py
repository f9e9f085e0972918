#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs per kernel and counter (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
wanted = sys.argv[2] if len(sys.argv) > 2 else "k_step"   # substring of the kernel names to report
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            key = (name, row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
kernels = sorted({k[0] for k in acc})
for kn in kernels:
    if wanted not in kn:
        continue
    print(kn)
    for (name, ctr), (tot, cnt) in sorted(acc.items()):
        if name == kn:
            print("  %-40s %18.1f  (mean of %d dispatches)" % (ctr, tot / cnt, cnt))
