#!/bin/bash
# round 4: re-sweep of the FullyObs time split after this round's changes to the step core
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4lava; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
B="timeout 120 python bench.py --workload lavacrossing_full --no-cpu-baseline --steps 2048 --warmup 256"
for nw in 1 2 3 4; do MG_ROLL_NW=$nw $B 2>&1 | line "lavacrossing_full NW=$nw "; done | tee $OUT/lava_sweep.txt
for r in 0.05 0.2 0.35; do MG_ROLL_RATIO=$r $B 2>&1 | line "lavacrossing_full NW=2 ratio=$r "; done | tee -a $OUT/lava_sweep.txt
for r in 0.2 0.35; do MG_ROLL_NW=3 MG_ROLL_RATIO=$r $B 2>&1 | line "lavacrossing_full NW=3 ratio=$r "; done | tee -a $OUT/lava_sweep.txt
MG_NT_BYTES=-1 $B 2>&1 | line "lavacrossing_full no NT stores " | tee -a $OUT/lava_sweep.txt
