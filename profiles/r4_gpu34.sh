#!/bin/bash
# round 4: DynamicObstacles in the loop, second cut (straight-line placement loop, codes staged by the dynamics wave: one grid copy)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4dyn2; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], (d['host_ms']-d['event_ms'])*1e3))"; }
timeout 600 python -m pytest tests/test_gpu_dynobs.py -q -p no:cacheprovider > $OUT/pytest_dynobs.log 2>&1; echo "dynobs tests rc=$?" | tee $OUT/rc.txt
tail -30 $OUT/pytest_dynobs.log
B="timeout 120 python bench.py --workload dynobs16x16 --no-cpu-baseline"
$B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop (default NW=3) " | tee $OUT/dynobs_bench.txt
MG_ROLL_NW=2 $B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop NW=2          " | tee -a $OUT/dynobs_bench.txt
MG_ROLL_NW=4 $B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop NW=4          " | tee -a $OUT/dynobs_bench.txt
MG_ROLL_SPLIT=0 $B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop time split     " | tee -a $OUT/dynobs_bench.txt
$B --fused 0 --steps 512 --warmup 64 2>&1 | line "dynobs16x16 x 65536 in-loop, one-step launches" | tee -a $OUT/dynobs_bench.txt
for n in 16384 32768 131072 262144; do $B --envs-per-gpu $n --steps 1024 --warmup 128 2>&1 | line "dynobs16x16 x $n "; done | tee -a $OUT/dynobs_bench.txt
