#!/bin/bash
# round 4, call 18: k_refill_lane (one lane per episode): the whole GPU suite, then the reset-heavy BASELINE configs with and without it
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4r; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3, d['config']['episodes_finished_rank0']))"; }
for rep in 1 2; do
for w in gotoredball lavacrossing_full doorkey8x8; do
  timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w lane refill"
  MG_LANE_GEN=0 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w wave refill"
done
done | tee $OUT/lane_refill.txt
