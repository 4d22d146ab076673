#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4v; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
for w in gotoredball lavacrossing_full doorkey8x8; do for wps in 1 2 3; do
  MG_LANE_WPS=$wps timeout 100 python bench.py --workload $w --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "$w wps=$wps"
done; done | tee $OUT/lane_refill_final.txt
MG_SPARE_RING=256 timeout 100 python bench.py --workload doorkey8x8 --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "doorkey8x8 R=256" | tee -a $OUT/lane_refill_final.txt
timeout 100 python bench.py --workload gotoredball --envs-per-gpu 262144 --steps 1024 --warmup 256 --no-cpu-baseline 2>&1 | line "gotoredball 262144 envs" | tee -a $OUT/lane_refill_final.txt
timeout 600 python -m pytest tests/test_gpu_fused_full.py tests/test_gpu_launch_lengths.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest.log
