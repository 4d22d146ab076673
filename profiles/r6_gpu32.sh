#!/bin/bash
# Round 6, call 32: what the FullyObs dynamics wave's staging copy (its 64 x 81-byte image stream into the ring, two LDS syncs, agent mark patch) costs:
# attribution build, MG_EXP 8192 = no staging copy (the encode wave encodes stale stagings: timing only), 2 = no encode, 64 = no resets
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 8192 8194 8256 8258 2 64; do
  MG_EXP=$x python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full MG_EXP=$x"
done | tee $OUT/attribution_fullyobs_staging_copy.txt
for x in 0 8192; do
  MG_ROLL_NW=3 MG_EXP=$x python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full MG_ROLL_NW=3 MG_EXP=$x"
  MG_EXP=$x python bench.py --workload doorkey8x8 --obs-mode full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 FullyObs MG_EXP=$x"
done | tee -a $OUT/attribution_fullyobs_staging_copy.txt
