#!/bin/bash
# Round 6, call 54: process_vis' bit extraction trimmed (3 of ~37 instructions per row: mg_roll.h MG_VIS_TRIM) -- the GPU suite, then the product against -DMG_VIS_TRIM=0
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call54.log; tail -3 $OUT/pytest_gpu_call54.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do for w in doorkey8x8 keycorridor gotoredball unlockpickup; do for lib in libminigrid_hip_novistrim.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $lib"
done; done; done | tee $OUT/ab_vis_trim.txt
