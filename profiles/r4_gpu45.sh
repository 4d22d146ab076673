#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4grb; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for w in gotoredball doorkey8x8 lavacrossing_full; do
  for sh in 1 2; do MG_ROLL_SHADOWS=$sh timeout 120 python bench.py --workload $w --no-cpu-baseline --steps 2048 --warmup 256 2>&1 | line "$w shadows=$sh "; done
done | tee $OUT/shadows2.txt
for f in 16 8; do MG_MAX_FUSED=$f timeout 120 python bench.py --workload gotoredball --no-cpu-baseline --steps 2048 --warmup 256 2>&1 | line "gotoredball max_fused=$f "; done | tee -a $OUT/shadows2.txt
