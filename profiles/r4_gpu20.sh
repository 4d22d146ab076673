#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4t; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
for w in gotoredball lavacrossing_full doorkey8x8; do
  for wps in 1 4; do MG_LANE_WPS=$wps timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w lane refill wps=$wps"; done
done | tee $OUT/lane_refill2.txt
cd /tmp; MG_LANE_WPS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o g -- python $ROOT/bench.py --workload gotoredball --steps 512 --warmup 128 --no-cpu-baseline > $OUT/prof.log 2>&1
head -5 $(find $OUT/prof -name '*kernel_stats.csv' | head -1) | cut -c1-160
python - $(find $OUT/prof -name '*kernel_trace.csv' | head -1) <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows[-24:]:
    print("%-40s start %9.1f us  dur %7.1f us  grid %s wg %s lds %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X","?"), r.get("Workgroup_Size_X","?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v","?"))))
PY
rm -rf $OUT/prof
