#!/bin/bash
# Round 6, call 40: two staged spare sets cost a GoToRedBall workgroup 5.4 KB of LDS (43 instead of 37.6 KB: three instead of four workgroups per CU) -- does the
# default (two sets for episodes of at most 64 steps) hurt the batches that need four slots per CU?
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for n in 16384 32768 49152 65536 131072; do for v in "MG_X=0" "MG_ROLL_SHADOWS=1" "MG_X=0" "MG_ROLL_SHADOWS=1"; do
  env $v python bench.py --workload gotoredball --envs-per-gpu $n --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball x $n $v"
done; done | tee $OUT/ab_shadow_sets_by_batch_size.txt
