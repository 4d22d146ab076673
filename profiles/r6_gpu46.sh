#!/bin/bash
# Round 6, call 46: VALU instructions of the view (MG_EXP 4 = no view codes) per workgroup-step: is the doubled count of call 45 the ROOMGRID kernel's or the level's?
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
cd /tmp
: > $OUT/sq_view_by_level.txt
for cfg in "doorkey8x8 --envs-per-gpu 32768 --max-steps 4096" "keycorridor --envs-per-gpu 32768 --max-steps 4096" "keycorridor --env-id MiniGrid-Unlock-v0 --envs-per-gpu 32768 --max-steps 4096" "gotoredball --env-id BabyAI-GoToObj-v0 --max-steps 4096" "gotoredball --env-id BabyAI-GoToRedBallNoDists-v0 --max-steps 4096"; do for x in 0 4; do
  echo "== $cfg MG_EXP=$x" >> $OUT/sq_view_by_level.txt
  rm -rf /tmp/sqx; MG_EXP=$x timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d /tmp/sqx -o x -- python $ROOT/bench.py --workload $cfg --steps 512 --warmup 128 --no-cpu-baseline > /tmp/sqx.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find /tmp/sqx -name '*counter_collection.csv' | head -1) | grep "k_roll7<[0-9], false, true" | awk -F, '{print $1, $NF}' >> $OUT/sq_view_by_level.txt
done; done
cat $OUT/sq_view_by_level.txt
