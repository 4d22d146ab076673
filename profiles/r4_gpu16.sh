#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4p; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3, d['config']['episodes_finished_rank0']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 46 62 110 174 238; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --workload gotoredball --steps 1024 --warmup 256 --no-cpu-baseline 2>&1 | line "gotoredball attr MG_EXP=$x "; done | tee $OUT/gotoredball_attr2.txt
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o g -- python $ROOT/bench.py --workload gotoredball --steps 512 --warmup 128 --no-cpu-baseline > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_gotoredball.csv; head -5 $OUT/kernel_stats_gotoredball.csv | cut -c1-150
python - $(find $OUT/prof -name '*kernel_trace.csv' | head -1) <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    print("%-40s start %9.1f us  dur %7.1f us  stream %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Stream_Id", r.get("Queue_Id","?"))))
PY
rm -rf $OUT/prof
