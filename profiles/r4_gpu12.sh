#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py tests/test_gpu_fused.py tests/test_gpu_fused_full.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_nt.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))"; }
for rep in 1 2; do
  for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball bosslevel; do
    timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w "
  done
  MG_ROLL_SPLIT=0 timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 time split "
  MG_ROLL_SPLIT=0 timeout 100 python bench.py --workload empty8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty8x8 time split "
  MG_ROLL_NW=3 timeout 100 python bench.py --workload lavacrossing_full --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full nw=3 "
  timeout 100 python bench.py --fused 0 --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | line "empty8x8 one launch per step "
  timeout 100 python bench.py --fused 0 --workload doorkey8x8 --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 one launch per step "
  timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized "
done | tee $OUT/bench_lines.txt
