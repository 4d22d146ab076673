#!/usr/bin/env python3
"""Per-kernel mean of a rocprofv3 --pmc counter_collection.csv (one counter per pass)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
for (name, ctr), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{ctr},{name},calls={n},mean={tot / n:.3f},total={tot:.1f}")
