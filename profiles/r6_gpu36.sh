#!/bin/bash
# Round 6, call 36: FullyObs with two / three waves per workgroup on the build with line-aligned rounds (the stream was what capped three waves before)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do for nw in 2 3 4; do
  MG_ROLL_NW=$nw python bench.py --workload lavacrossing_full --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full MG_ROLL_NW=$nw"
done; done | tee $OUT/ab_fullyobs_waves_after_alignment.txt
for nw in 2 3; do
  MG_ROLL_NW=$nw python bench.py --workload doorkey8x8 --obs-mode full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 FullyObs MG_ROLL_NW=$nw"
  MG_ROLL_NW=$nw python bench.py --workload empty8x8 --obs-mode full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "empty8x8 FullyObs MG_ROLL_NW=$nw"
  MG_ROLL_NW=$nw python bench.py --workload lavacrossing_full --envs-per-gpu 65536 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full x 65536 MG_ROLL_NW=$nw"
done | tee -a $OUT/ab_fullyobs_waves_after_alignment.txt
