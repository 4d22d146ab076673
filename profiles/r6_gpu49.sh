#!/bin/bash
# Round 6, call 49: k_roll7<GG_PICKUP> (RULE_PICKUP by itself: mg_step_pickup.hip) -- the GPU suite, then the product against -DMG_GOTO_TU=0 (both rule instantiations off)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call49.log; tail -3 $OUT/pytest_gpu_call49.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do for w in keycorridor unlockpickup blockedunlockpickup; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $lib"
done; done; done | tee $OUT/ab_pickup_tu.txt
for id in MiniGrid-ObstructedMaze-2Dlhb-v0 MiniGrid-ObstructedMaze-Full-v0 BabyAI-KeyCorridorS4R3-v0; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id $id --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 131072 $lib"
done; done | tee -a $OUT/ab_pickup_tu.txt
