#!/bin/bash
# Round 6, call 8: the scalar record of a step stored by the step's ENCODE wave (log split); GPU suite; the four BASELINE workloads + driver shape
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call8.log | tail -8
for k in 1 2; do
for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full keycorridor unlockpickup; do
  python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w steps 2048"
done; done | tee $OUT/bench_lines_baseline_call8.txt
for k in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver shape run $k"; done | tee -a $OUT/bench_lines_baseline_call8.txt
