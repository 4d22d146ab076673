#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4f; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_split.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  for r in 0 1 2 3 9 10 11; do
    MG_ROLL_DROT=$r timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty8x8 drot=$r  "
  done
  for r in 0 1 3 9 11 13; do
    MG_ROLL_DROT=$r timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 drot=$r  "
  done
done | tee $OUT/split_rotation.txt
