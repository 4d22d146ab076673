#!/bin/bash
# Round 6, call 44: instruction counters of the step kernel, BabyAI-GoToRedBall (max_steps 4096: rare episode ends; and the level's own 64) against Empty-8x8, x 32 768
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
cd /tmp
: > $OUT/sq_gotoredball_vs_empty_32768.txt
for cfg in "gotoredball --max-steps 4096" "gotoredball" "empty8x8 --envs-per-gpu 32768"; do
  echo "== $cfg" >> $OUT/sq_gotoredball_vs_empty_32768.txt
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_IFETCH"; do
    rm -rf /tmp/sqx; timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sqx -o x -- python $ROOT/bench.py --workload $cfg --steps 512 --warmup 128 --no-cpu-baseline > /tmp/sqx.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find /tmp/sqx -name '*counter_collection.csv' | head -1) | grep "k_roll7<[0-9], false, true" | cut -c1-170 >> $OUT/sq_gotoredball_vs_empty_32768.txt
  done
done
cat $OUT/sq_gotoredball_vs_empty_32768.txt
