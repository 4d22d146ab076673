#!/bin/bash
# Round 6, call 48: would RULE_PICKUP (KeyCorridor, UnlockPickup, BlockedUnlockPickup, ObstructedMaze) gain from an instantiation of its own like RULE_GOTO did?
# a build of mg_step_roomgrid.hip with the rule a compile-time constant (-DMG_FIXED_RULE=5) against the product
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do for w in keycorridor unlockpickup blockedunlockpickup; do for lib in libminigrid_hip.so libminigrid_hip_rulepickup.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $lib"
done; done; done | tee $OUT/ab_fixed_rule_pickup.txt
