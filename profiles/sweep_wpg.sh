#!/bin/bash
# tuning aid: steps/s of each BASELINE workload for every waves-per-group setting of k_step
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  for g in 4 2 1; do
    MG_WPG=$g python bench.py --workload $w --steps 1500 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w wpg=$g', '%.3f G steps/s'%(d['value']/1e9), '%.2f us/step'%(d['ms_per_step']*1e3))"
  done
done
