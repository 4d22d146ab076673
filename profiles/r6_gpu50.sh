#!/bin/bash
# Round 6, call 50: does k_roll7<GG_NONE> pay for the two rules it carries besides "none" (RULE_DYNOBS for the non-in-loop paths, RULE_SENTENCE's flag)?
# mg_step_none.hip with the rule a compile-time constant (-DMG_FIXED_RULE=0) against the product: the headline, DoorKey, LavaCrossing FullyObs
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do for w in empty8x8 doorkey8x8 lavacrossing_full; do for lib in libminigrid_hip.so libminigrid_hip_rulenone.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $lib"
done; done; done | tee $OUT/ab_fixed_rule_none.txt
