#!/bin/bash
# round 4: what bounds LavaCrossing FullyObs after the staged-copy split; BossLevel's generator tail (ring depth, waves per request segment)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 2 46 110 64; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 120 python bench.py --workload lavacrossing_full --no-cpu-baseline --steps 2048 --warmup 256 2>&1 | line "lavacrossing_full attr MG_EXP=$x "; done | tee $OUT/lava_attr_split.txt
B="timeout 200 python bench.py --workload bosslevel --no-cpu-baseline --steps 1024 --warmup 128"
$B 2>&1 | line "bosslevel x 131072 default (R=64, wps 16) " | tee $OUT/bosslevel_generator.txt
MG_SPARE_RING=128 $B 2>&1 | line "bosslevel x 131072 R=128 " | tee -a $OUT/bosslevel_generator.txt
MG_SPARE_RING=32 $B 2>&1 | line "bosslevel x 131072 R=32 " | tee -a $OUT/bosslevel_generator.txt
MG_REFILL_WPS=32 $B 2>&1 | line "bosslevel x 131072 wps 32 " | tee -a $OUT/bosslevel_generator.txt
MG_REFILL_WPS=8 $B 2>&1 | line "bosslevel x 131072 wps 8 " | tee -a $OUT/bosslevel_generator.txt
