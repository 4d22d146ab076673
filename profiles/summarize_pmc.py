#!/usr/bin/env python3
"""Per-kernel mean / total / per-call maximum of a rocprofv3 --pmc counter_collection.csv (one counter per pass).  The maximum is the
full fused launch of a run that also holds one-step reset observations of the same kernel."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0, 0.0])
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        v = float(row["Counter_Value"])
        acc[k][0] += 1
        acc[k][1] += v
        acc[k][2] = max(acc[k][2], v)
for (name, ctr), (n, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{ctr},{name},calls={n},mean={tot / n:.3f},total={tot:.1f},max={mx:.3f}")
