#!/bin/bash
# round 4: BossLevel -- where does a step go?  (attribution build: 2048 = no verifier, 4 = no codes, 2 = no observation stores, 16 = no transition)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4boss; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for n in 131072 32768; do
for x in 0 2048 2054 2070 6; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 200 python bench.py --workload bosslevel --envs-per-gpu $n --no-cpu-baseline --steps 512 --warmup 128 2>&1 | line "bosslevel x $n attr MG_EXP=$x "; done
done | tee $OUT/bosslevel_attr.txt
timeout 200 python bench.py --workload bosslevel --no-cpu-baseline --steps 512 --warmup 128 2>&1 | line "bosslevel x 131072 product " | tee -a $OUT/bosslevel_attr.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o x -- python $ROOT/bench.py --workload bosslevel --steps 256 --warmup 64 --no-cpu-baseline > $OUT/kt.log 2>&1
head -8 $(find $OUT/kt -name '*kernel_stats.csv' | head -1) | cut -c1-160 | tee $OUT/kernel_stats_bosslevel.txt
rm -rf $OUT/kt
