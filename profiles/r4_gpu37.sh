#!/bin/bash
# round 4: DynamicObstacles in the loop, third cut = the defaults (128 VGPRs, dynamics wave 0 + two encode waves over one grid copy)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4dyn5; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], (d['host_ms']-d['event_ms'])*1e3))"; }
timeout 600 python -m pytest tests/test_gpu_dynobs.py -q -p no:cacheprovider > $OUT/pytest_dynobs.log 2>&1; echo "dynobs tests rc=$?" | tee $OUT/rc.txt
tail -5 $OUT/pytest_dynobs.log
timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_roll.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "Dynamic or dynamic or same_step or wrapping or pickling" > $OUT/pytest_dynobs_old.log 2>&1; echo "older dynobs tests rc=$?" | tee -a $OUT/rc.txt
tail -3 $OUT/pytest_dynobs_old.log
for w in dynobs16x16 dynobs8x8 dynobs6x6; do
  B="timeout 120 python bench.py --workload $w --no-cpu-baseline"
  $B --steps 2048 --warmup 256 2>&1 | line "$w x 65536 in-loop                   "
  MG_DYN_INLOOP=0 $B --steps 512 --warmup 64 2>&1 | line "$w x 65536 round-3 launches         "
  $B --fused 0 --steps 512 --warmup 64 2>&1 | line "$w x 65536 in-loop, one-step launches"
done | tee $OUT/dynobs_bench.txt
B="timeout 120 python bench.py --workload dynobs16x16 --no-cpu-baseline"
for n in 16384 32768 131072 262144; do $B --envs-per-gpu $n --steps 1024 --warmup 128 2>&1 | line "dynobs16x16 x $n "; done | tee -a $OUT/dynobs_bench.txt
$B --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/dynobs_bench_line_20.json
timeout 200 python bench.py --workload dynobs16x16 --steps 2048 --warmup 256 2>&1 | tail -1 > $OUT/dynobs_bench_line.json
