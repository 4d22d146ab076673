#!/bin/bash
# Round 6, call 29: from which grid size on the STAGED split (one grid copy per workgroup: dynamics wave + encode wave) beats private copies per wave
# (four waves up to ~100 cells, fewer as the copies grow, ONE wave at 16 x 16).  MG_STAGED_CELLS = staged above this many cells (a temporary switch)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
for id in MiniGrid-Empty-16x16-v0 MiniGrid-DoorKey-16x16-v0 BabyAI-KeyCorridorS6R3-v0 MiniGrid-ObstructedMaze-Full-v0 MiniGrid-MemoryS13-v0 MiniGrid-SimpleCrossingS11N5-v0 MiniGrid-LavaCrossingS11N5-v0 MiniGrid-MemoryS11-v0 MiniGrid-RedBlueDoors-8x8-v0 MiniGrid-KeyCorridorS4R3-v0 MiniGrid-KeyCorridorS5R3-v0 MiniGrid-LavaCrossingS9N1-v0 MiniGrid-ObstructedMaze-2Dlhb-v0; do
  for t in 256 255 168 120 99 80; do
    MG_STAGED_CELLS=$t python bench.py --workload keycorridor --env-id $id --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 131072 MG_STAGED_CELLS=$t"
  done
done | tee $OUT/ab_staged_threshold.txt
