#!/bin/bash
# round 4: generating wavefronts per request segment (k_refill, one wavefront per episode) for the other levels it serves
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for w in keycorridor multiroom babyai_goto unlockpickup; do
  for wps in 16 4 2; do MG_REFILL_WPS=$wps timeout 200 python bench.py --workload $w --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "$w wps $wps "; done
done | tee $OUT/refill_wps_other_levels.txt
