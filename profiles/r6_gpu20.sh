#!/bin/bash
# Round 6, call 20: anatomy of a one-step launch (Env.step) on the attribution build: MG_EXP 16 = no transition, 4 = no view codes, 2 = no encode + stores, 8 = no scalar stores
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in empty8x8 doorkey8x8; do
for x in 0 2 6 22 30 8; do
  MG_EXP=$x python bench.py --workload $w --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline --dephase 0 2>/dev/null | line "$w one launch per step MG_EXP=$x"
done; done | tee $OUT/attribution_one_step_call20.txt
for n in 4096 16384 32768; do
  MG_EXP=0 python bench.py --workload empty8x8 --envs-per-gpu $n --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline --dephase 0 2>/dev/null | line "empty8x8 x $n one launch per step"
  MG_EXP=30 python bench.py --workload empty8x8 --envs-per-gpu $n --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline --dephase 0 2>/dev/null | line "empty8x8 x $n one launch per step MG_EXP=30"
done | tee -a $OUT/attribution_one_step_call20.txt
