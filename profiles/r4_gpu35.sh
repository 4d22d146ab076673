#!/bin/bash
# round 4: DynamicObstacles in the loop -- how many dynamics waves per SIMD?  (workgroup size x waves per workgroup x register budget)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4dyn3; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
B="timeout 120 python bench.py --workload dynobs16x16 --no-cpu-baseline --steps 1024 --warmup 128"
for lib in "" dynwpe4; do
  [ -n "$lib" ] && export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_$lib.so
  for epw in 64 32; do for nw in 1 2 3; do
    MG_ROLL_EPW=$epw MG_ROLL_NW=$nw $B 2>&1 | line "dynobs16x16 x 65536 lib=${lib:-product} EPW=$epw NW=$nw "
  done; done
  MG_ROLL_EPW=32 MG_ROLL_NW=2 $B --envs-per-gpu 262144 2>&1 | line "dynobs16x16 x 262144 lib=${lib:-product} EPW=32 NW=2 "
  MG_ROLL_EPW=32 MG_ROLL_NW=1 $B --envs-per-gpu 262144 2>&1 | line "dynobs16x16 x 262144 lib=${lib:-product} EPW=32 NW=1 "
done | tee $OUT/dynobs_waves_sweep.txt
