#!/usr/bin/env python3
"""kernel-trace analysis: per-kernel duration percentiles and the idle gap between consecutive dispatches"""
import csv
import sys
import numpy as np
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
names = sorted({n for _, _, n in rows})
for n in names:
    d = np.array([e - s for s, e, k in rows if k == n]) / 1e3
    print(f"{n[:60]:60s} calls={len(d):5d} dur us: p10={np.percentile(d,10):.2f} p50={np.percentile(d,50):.2f} p90={np.percentile(d,90):.2f} max={d.max():.1f} mean={d.mean():.2f}")
gaps = np.array([rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]) / 1e3
tail = gaps[len(gaps) // 3:]
print(f"gap between consecutive kernels (last 2/3 of the run) us: p10={np.percentile(tail,10):.2f} p50={np.percentile(tail,50):.2f} p90={np.percentile(tail,90):.2f} mean={tail.mean():.2f}")
