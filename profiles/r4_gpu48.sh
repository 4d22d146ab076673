#!/bin/bash
# round 4: the sentence levels' refill geometry (generating wavefronts per request segment, ring depth)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
B="timeout 200 python bench.py --workload bosslevel --no-cpu-baseline --steps 1024 --warmup 128"
for w in 1 2 4 8; do MG_REFILL_WPS=$w $B 2>&1 | line "bosslevel x 131072 wps $w R=64 "; done | tee $OUT/bosslevel_generator2.txt
for w in 2 4 8; do MG_REFILL_WPS=$w MG_SPARE_RING=128 $B 2>&1 | line "bosslevel x 131072 wps $w R=128 "; done | tee -a $OUT/bosslevel_generator2.txt
for w in 2 4 16; do MG_REFILL_WPS=$w $B --envs-per-gpu 32768 2>&1 | line "bosslevel x 32768 wps $w R=64 "; done | tee -a $OUT/bosslevel_generator2.txt
