#!/bin/bash
# Round 6, call 27: LDS occupancy of the FullyObs kernel (shadow sets 0 / 1, 2-4 waves) and of GoToRedBall (shadow sets 1 / 2), on the product build
# (k_roll7<GG_ROOMGRID / GG_LIGHT / GG_ROOMS> at four waves per SIMD); the GPU suite on that build
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call27.log; tail -3 $OUT/pytest_gpu_call27.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
for v in "MG_X=0" "MG_ROLL_SHADOWS=0" "MG_ROLL_SHADOWS=0 MG_ROLL_NW=3" "MG_ROLL_SHADOWS=0 MG_ROLL_NW=4" "MG_ROLL_SHADOWS=1 MG_ROLL_NW=3" "MG_X=0" "MG_ROLL_SHADOWS=0"; do
  env $v python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full $v"
done | tee $OUT/ab_fullyobs_lds_occupancy.txt
for v in "MG_X=0" "MG_ROLL_SHADOWS=2" "MG_X=0" "MG_ROLL_SHADOWS=2" "MG_X=0" "MG_ROLL_SHADOWS=2"; do
  env $v python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball $v"
done | tee -a $OUT/ab_fullyobs_lds_occupancy.txt
for v in "MG_X=0" "MG_ROLL_SHADOWS=0" "MG_ROLL_SHADOWS=2"; do
  env $v python bench.py --workload doorkey8x8 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 $v"
  env $v python bench.py --workload empty8x8 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "empty8x8 $v"
done | tee -a $OUT/ab_fullyobs_lds_occupancy.txt
