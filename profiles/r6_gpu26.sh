#!/bin/bash
# Round 6, call 26: k_roll7<GG_LIGHT> / <GG_ROOMS> (7x7 view, not STAGED) at four waves per SIMD (-DMG_LR_WPE=4) against three; the product now has GG_ROOMGRID at four
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do
for id in MiniGrid-Fetch-8x8-N3-v0 MiniGrid-GoToDoor-8x8-v0 MiniGrid-RedBlueDoors-8x8-v0 MiniGrid-MemoryS11-v0 BabyAI-PickupDist-v0 BabyAI-PutNextLocal-v0 BabyAI-OpenRedDoor-v0 MiniGrid-LockedRoom-v0; do
  for lib in libminigrid_hip.so libminigrid_hip_lrwpe4.so; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id $id --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 131072 $lib"
  done
done; done | tee $OUT/ab_light_rooms_waves_per_simd.txt
