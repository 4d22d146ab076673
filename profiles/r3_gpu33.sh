#!/bin/bash
# round 3: last sanity run of the committed tree (after an experiment on the FullyObs loop was measured and dropped)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3ag; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
for rep in 1 2; do for w in empty8x8 lavacrossing_full; do timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w %.3f G %.2f us/step frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac']))"; done; done | tee $OUT/bench_last.txt
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-sized %.3f G %.2f us/step' % (d['value']/1e9, d['ms_per_step']*1e3))" | tee -a $OUT/bench_last.txt
