#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4g; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  for w in empty8x8 doorkey8x8; do
    timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w slot-major        "
    MG_TRAJ_BLOCK_EXPERIMENT=1 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w blocks            "
    MG_ROLL_SPLIT=0 MG_ROLL_RATIO=0.09 MG_TRAJ_BLOCK_EXPERIMENT=1 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w blocks, time split"
  done
  for x in 0 32 2 6; do
    MG_EXP=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "attr (split) MG_EXP=$x  "
  done
done | tee $OUT/ab_blocks.txt
for i in 1 2; do MG_TRAJ_BLOCK_EXPERIMENT=1 timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized blocks"; done | tee -a $OUT/ab_blocks.txt
