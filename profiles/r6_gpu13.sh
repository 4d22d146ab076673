#!/bin/bash
# Round 6, call 13: lanes per generating wavefront of MultiRoom's packed refill
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
for lpw in 64 32 16 8; do
  MG_LANE_BURST=1024 MG_LANE_LPW=$lpw python bench.py --workload multiroom --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "multiroom steps 1024 (de-phased) MG_LANE_BURST=1024 MG_LANE_LPW=$lpw"
  MG_LANE_BURST=1024 MG_LANE_LPW=$lpw python bench.py --workload multiroom --steps 1024 --warmup 128 --no-cpu-baseline --dephase 0 2>/dev/null | line "multiroom steps 1024 --dephase 0 MG_LANE_BURST=1024 MG_LANE_LPW=$lpw"
done | tee $OUT/bench_lines_multiroom_call13.txt
