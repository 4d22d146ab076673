#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_full.py tests/test_gpu_philox.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
for w in gotoredball lavacrossing_full doorkey8x8; do
  for wps in 4 8 16 32; do MG_LANE_WPS=$wps timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w lane refill wps=$wps"; done
  MG_LANE_GEN=0 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w wave refill"
done | tee $OUT/lane_refill_wps.txt
