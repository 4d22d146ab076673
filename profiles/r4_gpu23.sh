#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4w; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
for rep in 1 2; do
  timeout 100 python bench.py --workload gotoredball --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball 32-env workgroups"
  MG_ROLL_EPW=64 timeout 100 python bench.py --workload gotoredball --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball 64-env workgroups"
  for n in 16384 32768; do
    timeout 100 python bench.py --workload empty8x8 --envs-per-gpu $n --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty8x8 $n envs epw 32"
    MG_ROLL_EPW=64 timeout 100 python bench.py --workload empty8x8 --envs-per-gpu $n --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty8x8 $n envs epw 64"
    timeout 100 python bench.py --fused 0 --workload empty8x8 --envs-per-gpu $n --steps 512 --warmup 64 --no-cpu-baseline 2>&1 | line "empty8x8 $n envs one launch per step epw 32"
    MG_ROLL_EPW=64 timeout 100 python bench.py --fused 0 --workload empty8x8 --envs-per-gpu $n --steps 512 --warmup 64 --no-cpu-baseline 2>&1 | line "empty8x8 $n envs one launch per step epw 64"
  done
done | tee $OUT/epw32.txt
