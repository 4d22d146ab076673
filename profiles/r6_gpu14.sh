#!/bin/bash
# Round 6, call 14: what a reset costs BabyAI-GoToRedBall x 32 768: no resets at all (attribution build, MG_EXP=64), shadow spares per launch 0 / 1 / 2
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
for cfg in "MG_X=0" "MG_ROLL_SHADOWS=2" "MG_ROLL_SHADOWS=0" "MG_X=0" "MG_ROLL_SHADOWS=2"; do
  env $cfg python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball $cfg"
done | tee $OUT/ab_gotoredball_shadows.txt
for cfg in "MG_X=0" "MG_ROLL_SHADOWS=2"; do
  env $cfg python bench.py --workload lavacrossing_full --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full $cfg"
  env $cfg python bench.py --workload gotoredball --envs-per-gpu 65536 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball x 65536 $cfg"
done | tee -a $OUT/ab_gotoredball_shadows.txt
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 64 16; do
  MG_EXP=$x python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball attribution build MG_EXP=$x"
done | tee -a $OUT/ab_gotoredball_shadows.txt
