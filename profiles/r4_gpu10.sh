#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4j; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))"; }
for rep in 1 2 3; do
  for x in 0 1 2 3 4 5; do
    MG_LAYOUT_X=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_layout.so timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty layout x=$x (obs blocks=$((x&1)), scalars $((x>>1)))  "
  done
done | tee $OUT/ab_layout.txt
for x in 0 3; do MG_LAYOUT_X=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_layout.so timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "doorkey layout x=$x  "; done | tee -a $OUT/ab_layout.txt
