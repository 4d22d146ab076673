#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py tests/test_gpu_fused.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_nt.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
for rep in 1 2; do
  for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
    timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w "
  done
  timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized "
  MG_NT_BYTES=0 timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized always nt "
  timeout 100 python bench.py --steps 64 --warmup 32 --no-cpu-baseline 2>/dev/null | line "empty 64 steps "
done | tee $OUT/bench_lines.txt
python profiles/host_overhead.py | tee $OUT/host_overhead_driver_sized.txt
