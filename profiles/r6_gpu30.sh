#!/bin/bash
# Round 6, call 30: LavaCrossing FullyObs x 131 072 (two rounds of 1 024 two-wave workgroups) on the attribution build: who paces a step -- the dynamics wave
# (transition + stream follow + staging copy) or the ONE encode wave (20 rounds of lookups + 12-byte stores per step)?  MG_EXP: 2 = no encode + stores,
# 32 = encode without stores, 64 = no resets, 16 = no transition
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for nw in 2 3; do for x in 0 2 32 64 66 16 18; do
  MG_ROLL_NW=$nw MG_EXP=$x python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full MG_ROLL_NW=$nw MG_EXP=$x"
done; done | tee $OUT/attribution_fullyobs.txt
for n in 32768 65536 262144; do for x in 0 2; do
  MG_EXP=$x python bench.py --workload lavacrossing_full --envs-per-gpu $n --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full x $n MG_EXP=$x"
done; done | tee -a $OUT/attribution_fullyobs.txt
