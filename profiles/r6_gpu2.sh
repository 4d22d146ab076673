#!/bin/bash
# Round 6, call 2: the rebuilt connect_all (rooms_reach: find_reach as a dozen 64-bit operations, recomputed only after a door was added) + the
# lattice loop without its division: GPU suite, attribution again, what it buys; shadow-spare staging off for the big grids (MG_ROLL_SHADOWS=0);
# where the headline kernel's LDS bank conflicts come from (attribution build, MG_EXP bits: 4 = no view, 2 = no encode, 16 = no transition)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], d['config']['episodes_finished_rank0']))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call2.log | tail -8
VAR=$ROOT/minigrid_amd/libminigrid_hip_genattr.so
for spec in "goto BabyAI-GoTo-v0 131072 1024" "multiroom MiniGrid-MultiRoom-N6-v0 65536 512" "bosslevel BabyAI-BossLevel-v0 131072 1024"; do
  set -- $spec
  MINIGRID_AMD_LIB=$VAR timeout 300 python profiles/gen_attr.py $2 $3 $4 > $OUT/refill_attribution_$1_after.txt 2>&1
  grep -E "episodes|rollout" $OUT/refill_attribution_$1_after.txt
done
for w in babyai_goto bosslevel multiroom; do
  for cfg in "MG_X=0" "MG_ROLL_SHADOWS=0" "MG_LANE_BURST=0" "MG_LANE_BURST=0 MG_ROLL_SHADOWS=0"; do
    env $cfg python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 $cfg"
  done
done | tee $OUT/ab_connect_all_shadows.txt
for cfg in "MG_X=0" "MG_ROLL_SHADOWS=0"; do
  env $cfg python bench.py --workload babyai_goto --steps 2304 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 2304 $cfg"
  env $cfg python bench.py --workload keycorridor --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "keycorridor steps 1024 $cfg"
done | tee -a $OUT/ab_connect_all_shadows.txt
# LDS bank conflicts of the headline kernel by part (attribution build)
cd /tmp
ATTR=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for exp in 0 4 2 6 16 22; do
  MINIGRID_AMD_LIB=$ATTR MG_EXP=$exp timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/lds$exp -o x -- python $ROOT/bench.py --workload empty8x8 --steps 320 --warmup 64 --no-cpu-baseline > $OUT/lds$exp.log 2>&1
  echo "== MG_EXP=$exp"
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/lds$exp -name '*counter_collection.csv' | head -1) | grep -E "k_roll7<0, false, true" | cut -c1-140
  tail -1 $OUT/lds$exp.log | line "MG_EXP=$exp"
  rm -rf $OUT/lds$exp
done > $OUT/lds_conflicts_by_part_empty8x8.txt
cat $OUT/lds_conflicts_by_part_empty8x8.txt
cd $ROOT
