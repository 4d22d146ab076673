#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4m; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))"; }
for rep in 1 2 3; do
  for w in empty8x8 doorkey8x8; do
    timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w nt obs, plain scalars (product) "
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_ntscal.so timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w nt obs + nt scalars             "
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_nont.so timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w no nt                           "
  done
  timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized product "
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_nont.so timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized no nt   "
done | tee $OUT/ab_nt2.txt
