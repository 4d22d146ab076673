#!/bin/bash
# round 4: ring depth for the other wavefront-per-episode levels
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last3; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for r in 64 128 256; do MG_SPARE_RING=$r timeout 200 python bench.py --workload keycorridor --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "keycorridor R=$r "; done | tee $OUT/ring_other_levels.txt
MG_SPARE_RING=128 MG_REFILL_WPS=32 timeout 200 python bench.py --workload keycorridor --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "keycorridor R=128 wps 32 " | tee -a $OUT/ring_other_levels.txt
for r in 128 256; do MG_SPARE_RING=$r timeout 200 python bench.py --workload babyai_goto --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "babyai_goto R=$r "; done | tee -a $OUT/ring_other_levels.txt
for r in 128 256; do MG_SPARE_RING=$r timeout 200 python bench.py --workload multiroom --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "multiroom R=$r "; done | tee -a $OUT/ring_other_levels.txt
