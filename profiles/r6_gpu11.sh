#!/bin/bash
# Round 6, call 11: 32 envs per workgroup for the big grids (LDS carve-up now follows the workgroup's env count: twice the workgroups, 7 instead of 4 per CU)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
MG_ROLL_EPW=32 timeout 600 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused_full.py -q -m gpu -n 4 2>&1 | tail -2 | tee $OUT/pytest_gpu_call11_epw32.log
for w in babyai_goto bosslevel multiroom keycorridor; do
  for cfg in "MG_ROLL_EPW=64" "MG_ROLL_EPW=32"; do
    env $cfg python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 (de-phased) $cfg"
  done
done | tee $OUT/ab_epw32_big_grids.txt
for w in babyai_goto bosslevel; do
  for cfg in "MG_ROLL_EPW=64" "MG_ROLL_EPW=32"; do
    env $cfg python bench.py --workload $w --envs-per-gpu 32768 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w x 32768 steps 1024 (de-phased) $cfg"
  done
done | tee -a $OUT/ab_epw32_big_grids.txt
