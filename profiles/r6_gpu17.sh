#!/bin/bash
# Round 6, call 17: one generating wavefront per request segment; deeper ring
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
for w in gotoredball keycorridor; do
for cfg in "MG_LANE_WPS=1" "MG_LANE_WPS=2" "MG_LANE_WPS=1 MG_SPARE_RING=512" "MG_LANE_WPS=2 MG_SPARE_RING=512"; do
  env $cfg python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $cfg"
done; done | tee $OUT/ab_lane_wps1.txt
