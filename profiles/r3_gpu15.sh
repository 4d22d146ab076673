#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3o; mkdir -p $OUT
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %-20s n=%-7d %7.3f G steps/s %6.2f us/step spl %d" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["config"]["steps_per_launch"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
for r in 16 32 64 128; do MG_SPARE_RING=$r MG_TRAJ_SLOTS=32 timeout 300 python bench.py --workload bosslevel --steps 512 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json bosslevel_ring$r; done 2>&1 | tee $OUT/boss_ring.txt
MG_SPARE_RING=64 timeout 300 python bench.py --workload bosslevel --envs-per-gpu 131072 --steps 512 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json bosslevel_ring64_131072 | tee -a $OUT/boss_ring.txt
cd /tmp; MG_SPARE_RING=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_boss -o boss -- python $ROOT/bench.py --workload bosslevel --steps 512 --warmup 128 --no-cpu-baseline > $OUT/prof_boss.log 2>&1
cp $(find $OUT/prof_boss -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bosslevel_ring64.csv; rm -rf $OUT/prof_boss; head -5 $OUT/kernel_stats_bosslevel_ring64.csv | cut -c1-160
