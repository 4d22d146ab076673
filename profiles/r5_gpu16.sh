#!/bin/bash
# Round 5: SQ counters of the final headline kernel (separate --pmc passes) and the kernel trace of BabyAI-GoTo with the burst hybrid
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5p; mkdir -p $OUT
bash profiles/pmc_sq_r4.sh r5p empty8x8 final > /dev/null 2>&1
cat $OUT/sq_counters_empty8x8_final.txt | cut -c1-160 | head -40
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_goto -o g -- python $ROOT/bench.py --workload babyai_goto --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/prof_goto.log 2>&1
f=$(find $OUT/prof_goto -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-170 $f | head -9 | tee $OUT/kernel_stats_babyai_goto_burst_hybrid.txt
rm -rf $OUT/prof_goto
