#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4y; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))"; }
L=$ROOT/minigrid_amd/libminigrid_hip_layout.so
for rep in 1 2 3; do
for w in empty8x8 doorkey8x8; do
for x in 0 1; do MG_LAYOUT_X=$x MINIGRID_AMD_LIB=$L timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w obs blocks=$x "; done
done; done | tee $OUT/ab_blocks_final.txt
