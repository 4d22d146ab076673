#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4aa; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for rep in 1 2 3; do
for al in 256 65536 2097152 4194304; do MG_SLOT_ALIGN=$al timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty slot align $al"; done
done | tee $OUT/slot_align.txt
for al in 256 2097152; do MG_SLOT_ALIGN=$al timeout 100 python bench.py --workload doorkey8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "doorkey slot align $al"; done | tee -a $OUT/slot_align.txt
