#!/bin/bash
# round 4: where does a DynamicObstacles step go?  attribution build + SQ counters
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4dyn; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
B="timeout 120 python bench.py --workload dynobs16x16 --no-cpu-baseline --steps 1024 --warmup 128"
for x in 0 512 1024 1536 2 1538; do MG_EXP=$x MINIGRID_AMD_LIB=$A $B 2>&1 | line "dynobs16x16 attr MG_EXP=$x "; done | tee $OUT/dynobs_attr.txt
for w in dynobs16x16; do
  for n in 16384 32768 131072; do $B --envs-per-gpu $n 2>&1 | line "dynobs16x16 x $n "; done | tee -a $OUT/dynobs_attr.txt
done
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq$i -o x -- python $ROOT/bench.py --workload dynobs16x16 --steps 320 --warmup 64 --no-cpu-baseline > $OUT/sq$i.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq$i -name '*counter_collection.csv' | head -1) | grep -E "k_roll7|k_step"
  rm -rf $OUT/sq$i
done > $OUT/sq_counters_dynobs16x16.txt
cat $OUT/sq_counters_dynobs16x16.txt | head -20
