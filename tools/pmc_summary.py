#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: tools/pmc_summary.py <dir> [<dir> ...]
Prints kernel, counter, dispatches, total, per-dispatch average (values as rocprofv3 reports them)."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                key = (name, row["Counter_Name"])
                acc[key][0] += 1
                acc[key][1] += float(row["Counter_Value"])
print("%-60s %-14s %10s %16s %16s" % ("kernel", "counter", "dispatches", "total", "per_dispatch"))
for (name, ctr), (n, tot) in sorted(acc.items()):
    print("%-60s %-14s %10d %16.0f %16.1f" % (name[:60], ctr, n, tot, tot / n))
