#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4u; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
for R in 128 256; do for wps in 2 4 8; do
  MG_SPARE_RING=$R MG_LANE_WPS=$wps timeout 100 python bench.py --workload gotoredball --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball lane refill R=$R wps=$wps"
done; done | tee $OUT/lane_refill_ring.txt
MG_LANE_GEN=0 MG_SPARE_RING=256 timeout 100 python bench.py --workload gotoredball --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball wave refill R=256" | tee -a $OUT/lane_refill_ring.txt
MG_LANE_GEN=0 timeout 100 python bench.py --workload gotoredball --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball wave refill R=64 (r3 default)" | tee -a $OUT/lane_refill_ring.txt
for R in 128 256; do MG_SPARE_RING=$R MG_LANE_WPS=1 timeout 100 python bench.py --workload lavacrossing_full --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "lavacrossing_full lane refill R=$R wps=1"; done | tee -a $OUT/lane_refill_ring.txt
