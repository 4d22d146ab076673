#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4h; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty product          "
  for v in dprio1 dprio3; do MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_$v.so timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty $v          "; done
  for x in 0 8 32 40 46 62; do
    MG_EXP=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "attr (split) MG_EXP=$x  "
  done
  timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "doorkey product          "
  for v in dprio3; do MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_$v.so timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "doorkey $v          "; done
done | tee $OUT/ab_prio.txt
