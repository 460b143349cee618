#!/usr/bin/env python3
"""Per-kernel averages of every counter collected by tools/pmc_sq.sh.

    python tools/pmc_table.py DIR > DIR/summary.csv

Reads every *_counter_collection.csv under DIR; prints kernel,counter,dispatches,avg (value per
dispatch, summed over the instances rocprofv3 reports for one dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))       # (kernel, counter) -> dispatch id -> value
for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wt_" not in k and "wt64_" not in k:
            continue
        acc[(k, r["Counter_Name"])][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
print("kernel,counter,dispatches,avg_per_dispatch")
for (k, c), d in sorted(acc.items()):
    vals = list(d.values())
    print(f'"{k}",{c},{len(vals)},{sum(vals) / len(vals):.1f}')
