#!/bin/bash
# Round 6, call 9: the cooperative spare fetch for every rule group but the plain one (second resets of a launch: 16 pending 8 x 8 grids per load instruction)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call9.log | tail -8
for k in 1 2; do
for w in gotoredball lavacrossing_full keycorridor unlockpickup unlock blockedunlockpickup; do
  python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w steps 2048"
done; done | tee $OUT/bench_lines_call9.txt
for w in gotoredball lavacrossing_full keycorridor empty8x8 multiroom babyai_goto; do
  python bench.py --workload $w --fused 0 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | line "$w one launch per step"
done | tee -a $OUT/bench_lines_call9.txt
