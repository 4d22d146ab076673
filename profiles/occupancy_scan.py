#!/usr/bin/env python3
"""Round 6: resident workgroups per CU of the step kernel of a bench.py run, from the dispatch records of a `rocprofv3 --kernel-trace` pass
(LDS_Block_Size, VGPR_Count, Accum_VGPR_Count, Workgroup_Size, Grid_Size per dispatch): what limits residency (LDS 160 KB per CU, 512 VGPRs per SIMD lane,
i.e. floor(512 / VGPRs) waves per SIMD x 4 SIMDs) and how many ROUNDS of workgroups a launch is (grid / (256 CUs x resident)) -- 2.67 rounds run like 3.
    python profiles/occupancy_scan.py <kernel_trace.csv> <label>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_roll7" in r["Kernel_Name"] or "k_step<" in r["Kernel_Name"]]
label = sys.argv[2] if len(sys.argv) > 2 else ""
if not rows:
    print(label, "no step kernel in the trace"); sys.exit(0)
# the timed region's launches are the last ones: take the most frequent (kernel, grid) among the last 16 dispatches
tail = rows[-16:]
col = lambda r, *names: next((r[n] for n in names if n in r), None)
if col(rows[0], "Grid_Size", "Grid_Size_X") is None:
    print(label, "columns:", list(rows[0].keys())); sys.exit(0)
key = lambda r: (r["Kernel_Name"].split("(")[0].replace("void mg::", ""), col(r, "Grid_Size", "Grid_Size_X"), col(r, "Workgroup_Size", "Workgroup_Size_X"), r["LDS_Block_Size"], r["VGPR_Count"], r.get("Accum_VGPR_Count", "0"))
best = max(set(map(key, tail)), key=lambda k: sum(1 for r in tail if key(r) == k))
name, grid, wgs, lds, vgpr, agpr = best
grid, wgs, lds, vgpr, agpr = int(grid), int(wgs), int(lds), int(vgpr), int(agpr or 0)
nw = wgs // 64
nwg = grid // wgs
regs = max(vgpr + agpr, 1)
waves_simd = min(8, 512 // ((regs + 7) // 8 * 8))
by_vgpr = waves_simd * 4 // nw
by_lds = (160 * 1024) // lds if lds else 99
res = max(1, min(by_vgpr, by_lds, 32))
rounds = nwg / (256 * res)
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail if key(r) == best]
print("%-46s %-52s wgs %6d x %d waves  LDS %6d B  VGPR %3d  resident/CU: by VGPR %2d, by LDS %2d -> %2d   rounds %.2f   launch %.1f us" %
      (label, name[:52], nwg, nw, lds, regs, by_vgpr, by_lds, res, rounds, sum(dur) / len(dur) / 1e3))
