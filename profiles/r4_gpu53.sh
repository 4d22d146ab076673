#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last4; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
timeout 600 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider -k "KeyCorridor or Unlock or ObstructedMaze or Blocked or roomgrid or RoomGrid" > $OUT/pytest_roomgrid.log 2>&1; echo "tests rc=$?" | tee $OUT/rc.txt
tail -3 $OUT/pytest_roomgrid.log
for w in keycorridor unlockpickup; do timeout 200 python bench.py --workload $w --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "$w (ring 128) "; done | tee $OUT/ring128.txt
