#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py tests/test_gpu_fused.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3, d['config']['episodes_finished_rank0']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in gotoredball lavacrossing_full; do
  timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w product"
  MG_ROLL_SPLIT=0 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w time split"
  for x in 46 110; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --workload $w --steps 1024 --warmup 256 --no-cpu-baseline 2>&1 | line "$w attr MG_EXP=$x "; done
done | tee $OUT/reset_copy.txt
for w in empty8x8 doorkey8x8; do timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "$w product"; done | tee -a $OUT/reset_copy.txt
