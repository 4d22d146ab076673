#!/bin/bash
# Round 6, call 10: where a BossLevel step goes (attribution build, MG_EXP bits: 2048 = no verifier, 2 = no encode + stores, 4 = no view codes, 16 = no transition, 32 = encode without stores)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in bosslevel babyai_goto; do
for x in 0 2048 2 6 2054 16 32 64; do
  MG_EXP=$x python bench.py --workload $w --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | line "$w MG_EXP=$x"
done; done | tee $OUT/attribution_biggrid_call10.txt
for n in 32768 65536; do
  MG_EXP=0 python bench.py --workload bosslevel --envs-per-gpu $n --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | line "bosslevel x $n MG_EXP=0"
done | tee -a $OUT/attribution_biggrid_call10.txt
