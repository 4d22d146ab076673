#!/bin/bash
# Round 6, call 52: the one-rule units (GG_RULE: goto, pickup, gotobig, pickupdesc, fetch) -- the GPU suite, then the product against -DMG_GOTO_TU=0 (every level on its group's unit)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call52.log; tail -3 $OUT/pytest_gpu_call52.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do
  for w in gotoredball babyai_goto unlockpickup; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $lib"
  done; done
  for id in BabyAI-PickupDist-v0 BabyAI-Pickup-v0 MiniGrid-Fetch-8x8-N3-v0 BabyAI-GoToImpUnlock-v0; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id $id --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 131072 $lib"
  done; done
done | tee $OUT/ab_one_rule_units.txt
