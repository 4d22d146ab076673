#!/bin/bash
# SQ-level counters of the step kernel (separate rocprofv3 --pmc passes; no tracing domains besides --kernel-trace)
# usage: bash profiles/pmc_sq_r4.sh <tag> <workload> <label> [ENV=VAL ...]
TAG=$1; W=$2; LABEL=$3; shift; shift; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --workload $W --steps 320 --warmup 64 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- $CMD > $OUT/sq$i.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep -E "k_roll7|k_step"
  rm -rf $OUT/sq$i
done > $OUT/sq_counters_${W}_$LABEL.txt
cat $OUT/sq_counters_${W}_$LABEL.txt | awk -F, '{print $1, $2, $NF}' | head -30
cd $ROOT
