#!/bin/bash
# Round 6, call 39: the dynamics wave polls its consumers only when what it last saw no longer covers the entry it is about to reuse (MG_POLL_CACHE), against a poll in every step
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lds_protocol.py tests/test_gpu_fused_full.py tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_gpu_call39.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do for w in empty8x8 doorkey8x8 keycorridor gotoredball; do
  for lib in libminigrid_hip_nopollcache.so libminigrid_hip.so; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $lib"
  done
done; done | tee $OUT/ab_poll_cache.txt
