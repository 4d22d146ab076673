#!/bin/bash
# Round 6, call 22: why a GoToRedBall step at 32 768 envs (2 workgroups per CU) takes 2.0 us even without resets when an Empty-8x8 step takes 1.35:
# the attribution build, MG_EXP bits 64 = no resets, 128 = no GoTo rule, 2 = no encode + stores, 4 = no view codes, 16 = no transition
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in empty8x8 gotoredball; do
for x in 0 64 192 66 70 86 80; do
  MG_EXP=$x python bench.py --workload $w --envs-per-gpu 32768 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w x 32768 MG_EXP=$x"
done; done | tee $OUT/attribution_gotoredball_vs_empty_32768.txt
for s in 0 1 2; do
  MG_EXP=64 MG_ROLL_SHADOWS=$s python bench.py --workload gotoredball --envs-per-gpu 32768 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball x 32768 MG_EXP=64 MG_ROLL_SHADOWS=$s"
  MG_EXP=0 MG_ROLL_SHADOWS=$s python bench.py --workload gotoredball --envs-per-gpu 32768 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball x 32768 MG_EXP=0 MG_ROLL_SHADOWS=$s"
done | tee -a $OUT/attribution_gotoredball_vs_empty_32768.txt
for nw in 2 3 4; do
  MG_EXP=64 MG_ROLL_NW=$nw python bench.py --workload gotoredball --envs-per-gpu 32768 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball x 32768 MG_EXP=64 MG_ROLL_NW=$nw"
  MG_EXP=64 MG_ROLL_NW=$nw python bench.py --workload empty8x8 --envs-per-gpu 32768 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "empty8x8 x 32768 MG_EXP=64 MG_ROLL_NW=$nw"
done | tee -a $OUT/attribution_gotoredball_vs_empty_32768.txt
