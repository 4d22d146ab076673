#!/bin/bash
# Round 6, call 37: with line-aligned rounds in place, two decisions re-measured: nontemporal against plain observation stores (MG_NT_BYTES=-1: never nontemporal),
# and the step's scalar record stored by the step's encode wave (-DMG_SCAL_BY_ENCODE=1)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do for w in empty8x8 doorkey8x8 lavacrossing_full keycorridor; do
  python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w product"
  MG_NT_BYTES=-1 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w MG_NT_BYTES=-1 (plain stores)"
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_scalenc.so python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w scalars by the encode wave"
done; done | tee $OUT/ab_after_alignment_nt_and_scalars.txt
