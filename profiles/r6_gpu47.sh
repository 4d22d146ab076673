#!/bin/bash
# Round 6, call 47: k_roll7<GG_GOTO> (the GoTo rule without its group's other four rules: mg_step_goto.hip) -- the GPU suite, then the product against -DMG_GOTO_TU=0
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call47.log; tail -3 $OUT/pytest_gpu_call47.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do for n in 32768 65536; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload gotoredball --envs-per-gpu $n --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball x $n $lib"
done; done; done | tee $OUT/ab_goto_tu.txt
for id in BabyAI-GoToObj-v0 BabyAI-GoToLocal-v0 BabyAI-GoToRedBlueBall-v0; do for lib in libminigrid_hip_nogototu.so libminigrid_hip.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload gotoredball --env-id $id --envs-per-gpu 65536 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$id x 65536 $lib"
done; done | tee -a $OUT/ab_goto_tu.txt
MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip.so python bench.py --workload gotoredball --obs-mode full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball FullyObs product" | tee -a $OUT/ab_goto_tu.txt
MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_nogototu.so python bench.py --workload gotoredball --obs-mode full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball FullyObs MG_GOTO_TU=0" | tee -a $OUT/ab_goto_tu.txt
