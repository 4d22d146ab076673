#!/bin/bash
# Round 6, call 51: which other rules would gain from an instantiation of their own -- fixed-rule builds of their group's unit against the product:
# RULE_GOTO_BIG (BabyAI-GoTo, STAGED 22 x 22), RULE_PICKUPDESC (BabyAI-PickupDist, BabyAI-Pickup), RULE_FETCH (Fetch-8x8-N3)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do
  for lib in libminigrid_hip.so libminigrid_hip_rulerooms13.so; do MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload babyai_goto --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto $lib"; done
  for lib in libminigrid_hip.so libminigrid_hip_rulerooms10.so; do MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id BabyAI-PickupDist-v0 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "BabyAI-PickupDist x 131072 $lib"; done
  for lib in libminigrid_hip.so libminigrid_hip_rulerooms10.so; do MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id BabyAI-Pickup-v0 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "BabyAI-Pickup x 131072 $lib"; done
  for lib in libminigrid_hip.so libminigrid_hip_rulelight2.so; do MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id MiniGrid-Fetch-8x8-N3-v0 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "Fetch-8x8-N3 x 131072 $lib"; done
done | tee $OUT/ab_fixed_rule_others.txt
