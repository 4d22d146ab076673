#!/bin/bash
# Round 6, call 53: one-rule units for every rule (MG_ONE_RULE_UNITS, mg_launch.h) -- the GPU suite, then the nine new ones against -DMG_GOTO_TU=0 (every level on its group's unit)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call53.log; tail -3 $OUT/pytest_gpu_call53.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for id in MiniGrid-Unlock-v0 MiniGrid-GoToObject-8x8-N2-v0 MiniGrid-PutNear-8x8-N3-v0 BabyAI-OpenRedDoor-v0 BabyAI-Open-v0 BabyAI-PutNextLocal-v0 BabyAI-PutNextS7N4-v0 BabyAI-OpenDoor-v0 MiniGrid-GoToDoor-8x8-v0 MiniGrid-RedBlueDoors-8x8-v0 MiniGrid-MemoryS11-v0 MiniGrid-MemoryS17Random-v0; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload keycorridor --env-id $id --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 131072 $lib"
done; done | tee $OUT/ab_one_rule_units_all.txt
