#!/bin/bash
# Round 6, call 42: what carrying every rule of its group costs k_roll7<GG_ROOMGRID>: the product against a build whose rule is the compile-time constant RULE_GOTO
# (-DMG_FIXED_RULE=1, mg_step_roomgrid.hip only): BabyAI-GoToRedBall
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do for n in 32768 65536; do for lib in libminigrid_hip.so libminigrid_hip_rulegoto.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload gotoredball --envs-per-gpu $n --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball x $n $lib"
done; done; done | tee $OUT/ab_fixed_rule.txt
