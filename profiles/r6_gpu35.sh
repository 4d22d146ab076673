#!/bin/bash
# Round 6, call 35: line-aligned encode rounds (mg_roll.h encode_quads `shift`; the FullyObs loop): GPU suite, then the product against -DMG_ALIGN_ROUNDS=0 on one box
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call35.log; tail -3 $OUT/pytest_gpu_call35.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do
for w in empty8x8 doorkey8x8 lavacrossing_full keycorridor; do
  for lib in libminigrid_hip_noalign.so libminigrid_hip.so; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w $lib"
  done
done; done | tee $OUT/ab_line_aligned_rounds.txt
for lib in libminigrid_hip_noalign.so libminigrid_hip.so libminigrid_hip_noalign.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver-sized $lib"
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball $lib"
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload dynobs16x16 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "dynobs16x16 $lib"
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload bosslevel --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "bosslevel $lib"
done | tee -a $OUT/ab_line_aligned_rounds.txt
