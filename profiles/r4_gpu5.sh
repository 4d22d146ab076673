#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py tests/test_gpu_fused.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_split.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2 3; do
  for w in empty8x8 doorkey8x8; do
    timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w split       "
    MG_ROLL_SPLIT=0 MG_ROLL_RATIO=0.09 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w time split  "
  done
done | tee $OUT/ab_split.txt
for i in 1 2; do timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized "; done | tee -a $OUT/ab_split.txt
bash profiles/pmc_sq_r4.sh r4e empty8x8 split
