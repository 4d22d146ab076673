#!/bin/bash
# Round 6, call 18: ring slots per request and refill (MG_LANE_CAP; 0 = every free slot, as before)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
for w in gotoredball lavacrossing_full; do
for cfg in "MG_LANE_CAP=0" "MG_LANE_CAP=2" "MG_LANE_CAP=3" "MG_LANE_CAP=4" "MG_LANE_CAP=6" "MG_LANE_CAP=8"; do
  env $cfg python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $cfg"
done; done | tee $OUT/ab_lane_cap.txt
for w in doorkey8x8 keycorridor unlockpickup multiroom empty8x8; do
for cfg in "MG_LANE_CAP=0" "MG_LANE_CAP=4"; do
  env $cfg python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $cfg"
done; done | tee -a $OUT/ab_lane_cap.txt
