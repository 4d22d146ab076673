#!/bin/bash
# round 4, call 3: the SPLIT k_roll7 (one dynamics wave + encode waves): fused parity tests first, then A/B against the time split
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4c; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
timeout 900 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py tests/test_gpu_fused.py tests/test_gpu_fused_full.py -x -q -m gpu -n 4 2>&1 | tail -8 | tee $OUT/pytest_split.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  for w in empty8x8 doorkey8x8 gotoredball; do
    timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w split       "
    MG_ROLL_SPLIT=0 MG_ROLL_RATIO=0.09 timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w time split  "
    MG_ROLL_NW=3 timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w split nw=3  "
  done
done | tee $OUT/ab_split.txt
for i in 1 2; do timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized "; done | tee -a $OUT/ab_split.txt
