#!/bin/bash
# Round 6, call 41: the 9 x 9 levels (100 bytes per grid copy): four waves per workgroup = 48 KB of LDS = three workgroups per CU (2 048 workgroups: 2.67 rounds);
# three waves = 38.5 KB = four per CU (two rounds).  And 11 x 11 (three waves, 47 KB) against two.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for id in MiniGrid-LavaCrossingS9N1-v0 MiniGrid-SimpleCrossingS9N3-v0 MiniGrid-MemoryS9-v0; do for n in 65536 131072 262144; do for nw in 4 3 4 3; do
  MG_ROLL_NW=$nw python bench.py --workload keycorridor --env-id $id --envs-per-gpu $n --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x $n MG_ROLL_NW=$nw"
done; done; done | tee $OUT/ab_9x9_waves_per_workgroup.txt
for id in MiniGrid-SimpleCrossingS11N5-v0 MiniGrid-MemoryS11-v0; do for nw in 3 2 4; do
  MG_ROLL_NW=$nw python bench.py --workload keycorridor --env-id $id --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 131072 MG_ROLL_NW=$nw"
done; done | tee -a $OUT/ab_9x9_waves_per_workgroup.txt
