#!/bin/bash
# Round 6, call 3: speculative connect_all + vectorised 4-draw headers + LDS peeks + per-kind draw budgets; shadows off for the big grids by default;
# the device policy's actions staged by the whole workgroup (MG_ACT_STAGE A/B); the de-phased bench line in the driver's shape
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes %d, in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c['episodes_finished_rank0'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call3.log | tail -8
VAR=$ROOT/minigrid_amd/libminigrid_hip_genattr.so
for spec in "goto BabyAI-GoTo-v0 131072 1024" "multiroom MiniGrid-MultiRoom-N6-v0 65536 512" "bosslevel BabyAI-BossLevel-v0 131072 1024"; do
  set -- $spec
  MINIGRID_AMD_LIB=$VAR timeout 300 python profiles/gen_attr.py $2 $3 $4 > $OUT/refill_attribution_$1_after2.txt 2>&1
  grep -v amdgpu.ids $OUT/refill_attribution_$1_after2.txt
done
for w in babyai_goto bosslevel multiroom keycorridor unlockpickup; do
  for cfg in "MG_X=0" "MG_LANE_BURST=0" "MG_ROLL_STAGED=0" "MG_ROLL_NW=3" "MG_LANE_BURST=8192"; do
    env $cfg python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 $cfg"
  done
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline --dephase 0 2>/dev/null | line "$w steps 1024 --dephase 0"
done | tee $OUT/bench_lines_generators_call3.txt
python bench.py --workload babyai_goto --steps 2304 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 2304" | tee -a $OUT/bench_lines_generators_call3.txt
for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do
  for cfg in "MG_ACT_STAGE=1" "MG_ACT_STAGE=0" "MG_ACT_STAGE=1" "MG_ACT_STAGE=0"; do
    env $cfg python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w steps 2048 $cfg"
  done
done | tee $OUT/ab_action_staging.txt
for k in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | line "driver shape (dephased) run $k"; done | tee $OUT/bench_driver_shape_dephased.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --dephase 0 --no-cpu-baseline 2>/dev/null | line "driver shape --dephase 0" | tee -a $OUT/bench_driver_shape_dephased.txt
