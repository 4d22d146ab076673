#!/bin/bash
# Round 6, call 23: gen_goto_lane (the GoTo family's lane generator as one loop of draws): parity (the generator / parity / roll tests), then GoToRedBall
# x 32 768 and x 65 536 per-segment (2 waves per segment) and packed (64 / 32 / 16 busy lanes per wave)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call23.log; tail -3 $OUT/pytest_gpu_call23.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
for n in 32768 65536; do
  for v in "MG_X=0" "MG_X=0" "MG_LANE_PACKED=1" "MG_LANE_PACKED=1 MG_LANE_LPW=32" "MG_LANE_PACKED=1 MG_LANE_LPW=16" "MG_LANE_CAP=0" "MG_LANE_PACKED=1 MG_LANE_CAP=0" "MG_LANE_CAP=4"; do
    env $v python bench.py --workload gotoredball --envs-per-gpu $n --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball x $n $v"
  done
done | tee $OUT/ab_goto_lane_state_machine.txt
