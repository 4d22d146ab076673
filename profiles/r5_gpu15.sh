#!/bin/bash
# Round 5: the burst hybrid refill + the deeper default ring of the big-grid maze levels: GPU suite (default, then the maze / sentence tests with a
# threshold that makes bursts frequent), then what it buys
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5o; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], d['config']['episodes_finished_rank0']))
except Exception as ex: print('$1 FAILED', ex)"; }
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 | tee $OUT/pytest_gpu_full_suite.log
MG_LANE_BURST=48 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roll.py tests/test_gpu_fused.py tests/test_gpu_synths5r2.py -q -m gpu -n 4 2>&1 | tail -4 | tee $OUT/pytest_burst48.log
for w in babyai_goto bosslevel multiroom; do
  for cfg in "MG_LANE_BURST=0" "MG_LANE_BURST=32768" "MG_LANE_BURST=16384"; do
    env $cfg python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 $cfg"
  done
done | tee $OUT/ab_burst_hybrid.txt
for cfg in "MG_LANE_BURST=0" "MG_LANE_BURST=32768"; do
  env $cfg python bench.py --workload babyai_goto --steps 2304 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 2304 $cfg"
  env $cfg python bench.py --workload babyai_goto --steps 320 --warmup 64 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 320 $cfg"
done | tee -a $OUT/ab_burst_hybrid.txt
