#!/bin/bash
# round 4, call 9: packed per-env scalars (ABI 3) + split k_roll7 with a prioritised dynamics wave: the whole GPU suite, then bench lines
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4i; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
    timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w       "
  done
  for x in 0 8 32 40; do
    MG_EXP=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "attr MG_EXP=$x  "
  done
  timeout 100 python bench.py --fused 0 --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | line "empty8x8 one launch per step "
  timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized "
done | tee $OUT/bench_lines.txt
