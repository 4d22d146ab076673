#!/bin/bash
# round 4: BossLevel after the verifier restatement: records in LDS vs global (A/B build), attribution of what is left
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4boss4; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
B="timeout 200 python bench.py --workload bosslevel --no-cpu-baseline --steps 512 --warmup 128"
for n in 131072 32768; do
  $B --envs-per-gpu $n 2>&1 | line "bosslevel x $n records in LDS    "
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_instrglobal.so $B --envs-per-gpu $n 2>&1 | line "bosslevel x $n records in global "
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_instrglobal.so MG_ROLL_SHADOWS=1 $B --envs-per-gpu $n 2>&1 | line "bosslevel x $n records in global, shadows on "
done | tee $OUT/bosslevel_records_ab.txt
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 2048 2054 2070 6; do MG_EXP=$x MINIGRID_AMD_LIB=$A $B 2>&1 | line "bosslevel x 131072 attr MG_EXP=$x "; done | tee $OUT/bosslevel_attr2.txt
