#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4x; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for rep in 1 2; do
timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty product"
for x in 0 8 32 40 44 46 62; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty attr MG_EXP=$x "; done
done | tee $OUT/attribution_final.txt
for x in 0 8 32 40 46; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "doorkey attr MG_EXP=$x "; done | tee -a $OUT/attribution_final.txt
