#!/bin/bash
# Round 6, call 38: the headline on the aligned build, taken apart once more (attribution build): MG_EXP 8 = no scalar stores, 32 = encode without stores,
# 256 = observation stores into an L2-resident window (same instructions, no HBM stream), 2 = no encode + stores, 6 = no view either, 64 = no resets
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in empty8x8 doorkey8x8; do for x in 0 8 32 40 256 264 2 6 64; do
  MG_EXP=$x python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w MG_EXP=$x"
done; done | tee $OUT/attribution_headline_aligned_build.txt
