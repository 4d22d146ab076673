#!/bin/bash
# Round 6, call 25: (1) babyai_done_actions="enum" on the GPU; (2) k_roll7<GG_ROOMGRID> at four waves per SIMD (128 VGPRs, -DMG_RG_WPE=4) against three (151)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -n 4 -k "done_actions" 2>&1 | tail -5 | tee $OUT/pytest_gpu_call25_done_enum.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2; do
for w in keycorridor unlock unlockpickup blockedunlockpickup gotoredball; do
  for lib in libminigrid_hip.so libminigrid_hip_rgwpe4.so; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $lib"
  done
done; done | tee $OUT/ab_roomgrid_waves_per_simd.txt
for lib in libminigrid_hip.so libminigrid_hip_rgwpe4.so; do
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload gotoredball --envs-per-gpu 65536 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball x 65536 $lib"
  MINIGRID_AMD_LIB=$ROOT/minigrid_amd/$lib python bench.py --workload gotoredball --envs-per-gpu 131072 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "gotoredball x 131072 $lib"
done | tee -a $OUT/ab_roomgrid_waves_per_simd.txt
