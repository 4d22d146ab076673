#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4dyn3; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for w in dynobs16x16 dynobs8x8 dynobs6x6; do
B="timeout 120 python bench.py --workload $w --no-cpu-baseline --steps 1024 --warmup 128"
for lib in "" dynwpe4; do
  if [ -n "$lib" ]; then export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_$lib.so; else unset MINIGRID_AMD_LIB; fi
  for nw in 1 3 4; do
    MG_ROLL_NW=$nw $B 2>&1 | line "$w x 65536 lib=${lib:-product} NW=$nw "
  done
  for drot in 0 1 3; do MG_ROLL_NW=3 MG_ROLL_DROT=$drot $B 2>&1 | line "$w x 65536 lib=${lib:-product} NW=3 DROT=$drot "; done
done
done | tee $OUT/dynobs_waves_sweep2.txt
