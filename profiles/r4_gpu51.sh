#!/bin/bash
# round 4: does the generator stream still need to outrank the step stream?  (measured in round 2 with a wavefront per episode and a ring of 32)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last3; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for w in gotoredball doorkey8x8 lavacrossing_full bosslevel keycorridor; do
  for p in 1 0; do MG_GEN_PRIO=$p timeout 200 python bench.py --workload $w --no-cpu-baseline --steps 2048 --warmup 256 2>&1 | line "$w generator priority $p "; done
done | tee $OUT/gen_priority.txt
