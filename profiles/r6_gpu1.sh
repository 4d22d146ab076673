#!/bin/bash
# Round 6, call 1: per-phase attribution of the wavefront-per-episode generators (the -DMG_GEN_ATTR variant library), SQ counters and kernel traces of
# k_refill for the three generator-bound families ("no per-phase attribution of k_refill exists").
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
VAR=$ROOT/minigrid_amd/libminigrid_hip_genattr.so
for spec in "goto BabyAI-GoTo-v0 131072 1024" "multiroom MiniGrid-MultiRoom-N6-v0 65536 512" "bosslevel BabyAI-BossLevel-v0 131072 1024"; do
  set -- $spec
  MINIGRID_AMD_LIB=$VAR timeout 300 python profiles/gen_attr.py $2 $3 $4 > $OUT/refill_attribution_$1.txt 2>&1
  tail -32 $OUT/refill_attribution_$1.txt
done
# SQ counters of the generator kernels (product library; cooperative kernels only: MG_LANE_BURST=0 MG_LANE_DIRECT=0), separate --pmc passes
cd /tmp
for W in babyai_goto multiroom bosslevel; do
  CMD="python $ROOT/bench.py --workload $W --steps 320 --warmup 64 --no-cpu-baseline"
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH"; do
    i=$((i+1))
    MG_LANE_BURST=0 MG_LANE_DIRECT=0 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- $CMD > $OUT/sq$i.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep -E "k_refill|k_generate|k_roll7"
    rm -rf $OUT/sq$i
  done > $OUT/sq_counters_generators_$W.txt
  cut -c1-200 $OUT/sq_counters_generators_$W.txt | head -40
  MG_LANE_BURST=0 MG_LANE_DIRECT=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o k -- $CMD > $OUT/kt_$W.log 2>&1
  f=$(find $OUT/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-200 $f | head -8 > $OUT/kernel_stats_cooperative_$W.txt
  cat $OUT/kernel_stats_cooperative_$W.txt; tail -1 $OUT/kt_$W.log | cut -c1-300
  rm -rf $OUT/kt
done
cd $ROOT
