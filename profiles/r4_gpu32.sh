#!/bin/bash
# round 4: DynamicObstacles inside the fused step kernel (k_roll7<GG_DYNOBS>): parity, then the workload at 65 536 x 16x16
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4dyn; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], (d['host_ms']-d['event_ms'])*1e3))"; }
timeout 600 python -m pytest tests/test_gpu_dynobs.py -q -p no:cacheprovider > $OUT/pytest_dynobs.log 2>&1; echo "dynobs tests rc=$?" | tee $OUT/rc.txt
tail -40 $OUT/pytest_dynobs.log
timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_roll.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "Dynamic or dynamic or same_step or wrapping or pickling" > $OUT/pytest_dynobs_old.log 2>&1; echo "older dynobs tests rc=$?" | tee -a $OUT/rc.txt
tail -8 $OUT/pytest_dynobs_old.log
B="timeout 120 python bench.py --workload dynobs16x16 --no-cpu-baseline"
$B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop (default)     " | tee $OUT/dynobs_bench.txt
MG_ROLL_NW=4 $B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop NW=4          " | tee -a $OUT/dynobs_bench.txt
MG_ROLL_EPW=32 MG_ROLL_NW=4 $B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop EPW=32 NW=4   " | tee -a $OUT/dynobs_bench.txt
MG_ROLL_SPLIT=0 $B --steps 2048 --warmup 256 2>&1 | line "dynobs16x16 x 65536 in-loop time split     " | tee -a $OUT/dynobs_bench.txt
MG_DYN_INLOOP=0 $B --steps 512 --warmup 64 2>&1 | line "dynobs16x16 x 65536 round-3 launches       " | tee -a $OUT/dynobs_bench.txt
$B --fused 0 --steps 512 --warmup 64 2>&1 | line "dynobs16x16 x 65536 in-loop, one-step launches" | tee -a $OUT/dynobs_bench.txt
$B --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/dynobs_bench_line_20.json
$B --steps 2048 --warmup 256 2>&1 | tail -1 > $OUT/dynobs_bench_line.json
