#!/bin/bash
# Round 6, call 19: the adaptive slot cap (a quarter of the free slots, at least two) over short and long windows; GPU suite
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call19.log | tail -8
for steps in 2048 16384; do
for cfg in "MG_LANE_CAP=0" "MG_X=0" "MG_LANE_CAP=3"; do
  env $cfg python bench.py --workload gotoredball --steps $steps --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball steps $steps $cfg"
done; done | tee $OUT/ab_lane_cap_adaptive.txt
for w in lavacrossing_full doorkey8x8 keycorridor empty8x8 multiroom; do
for cfg in "MG_LANE_CAP=0" "MG_X=0"; do
  env $cfg python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $cfg"
done; done | tee -a $OUT/ab_lane_cap_adaptive.txt
