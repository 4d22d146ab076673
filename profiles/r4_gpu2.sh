#!/bin/bash
# round 4, call 2: straight-line env_transition -- the whole GPU suite (it changes every level's dynamics), the extended write-stream
# microbenchmark (packed / far scalar layouts), bench lines with a split-ratio sweep and the MG_EXP attribution of the new build
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4b; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
timeout 1200 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rollstore.hip -o /tmp/rollstore && timeout 120 /tmp/rollstore | tee $OUT/rollstore2.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  for r in 0.12 0.09 0.06 0.04; do MG_ROLL_RATIO=$r timeout 100 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "empty ratio $r "; done
  for x in 0 32 2 6 22; do
    MG_EXP=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so timeout 100 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "attr MG_EXP=$x  "
  done
done | tee $OUT/ab_transition.txt
for w in doorkey8x8 lavacrossing_full gotoredball; do for r in 0.12 0.06; do
  MG_ROLL_RATIO=$r timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w ratio $r "; done; done | tee -a $OUT/ab_transition.txt
for i in 1 2; do timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized "; done | tee -a $OUT/ab_transition.txt
