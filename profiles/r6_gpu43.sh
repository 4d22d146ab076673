#!/bin/bash
# Round 6, call 43: BabyAI-GoToRedBall x 32 768 with max_steps = 4096 (an episode end in 0.3 % of the env-steps: the step itself), against Empty-8x8, attribution build:
# MG_EXP 128 = no GoTo rule, 2 = no encode + stores, 6 = no view either, 16 = no transition, 22 = neither
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 128 2 6 16 22; do
  MG_EXP=$x python bench.py --workload gotoredball --max-steps 4096 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball max_steps 4096 MG_EXP=$x"
  MG_EXP=$x python bench.py --workload empty8x8 --envs-per-gpu 32768 --max-steps 4096 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "empty8x8 x 32768 max_steps 4096 MG_EXP=$x"
done | tee $OUT/attribution_gotoredball_step_itself.txt
