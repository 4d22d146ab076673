#!/bin/bash
# Round 6, call 34: does the 64-byte misalignment of every other workgroup's observation block cost bandwidth?  MG_EXP 16384 = blocks moved down to a line start (timing only)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in lavacrossing_full empty8x8 doorkey8x8 gotoredball; do for x in 0 16384 0 16384; do
  MG_EXP=$x python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w MG_EXP=$x"
done; done | tee $OUT/attribution_store_alignment.txt
for x in 0 16384; do
  MG_ROLL_NW=3 MG_EXP=$x python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full MG_ROLL_NW=3 MG_EXP=$x"
done | tee -a $OUT/attribution_store_alignment.txt
