#!/bin/bash
# Round 6, call 45: instruction counts per part of the step, GoToRedBall (max_steps 4096) against Empty-8x8, x 32 768, attribution build: MG_EXP 0 / 6 (no view, no encode:
# the dynamics wave + the log followers) / 22 (no transition either) / 128 (no GoTo rule)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
export MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so
cd /tmp
: > $OUT/sq_by_part_gotoredball_vs_empty.txt
for cfg in "gotoredball --max-steps 4096" "empty8x8 --envs-per-gpu 32768 --max-steps 4096"; do for x in 0 6 22 128 134; do
  echo "== $cfg MG_EXP=$x" >> $OUT/sq_by_part_gotoredball_vs_empty.txt
  rm -rf /tmp/sqx; MG_EXP=$x timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /tmp/sqx -o x -- python $ROOT/bench.py --workload $cfg --steps 512 --warmup 128 --no-cpu-baseline > /tmp/sqx.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find /tmp/sqx -name '*counter_collection.csv' | head -1) | grep "k_roll7<[0-9], false, true" | awk -F, '{print $1, $NF}' >> $OUT/sq_by_part_gotoredball_vs_empty.txt
done; done
cat $OUT/sq_by_part_gotoredball_vs_empty.txt
