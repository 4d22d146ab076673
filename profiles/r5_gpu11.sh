#!/bin/bash
# Round 5: direct generation on lanes for EVERY level (reset(seed) and its ring fill), refills as before -- the GPU suite, then the ring-fill times
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5k; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 | tee $OUT/pytest_gpu_full_suite.log
for w in "BabyAI-GoTo-v0 131072" "BabyAI-BossLevel-v0 131072" "MiniGrid-MultiRoom-N6-v0 65536" "MiniGrid-ObstructedMaze-Full-v1 32768" "BabyAI-Pickup-v0 65536" "MiniGrid-MemoryS17Random-v0 65536"; do
  for d in 1 0; do MG_LANE_DIRECT=$d python profiles/reset_latency_lanes.py $w 2>/dev/null; done
done | tee $OUT/reset_latency_lanes.txt
for w in babyai_goto multiroom bosslevel keycorridor gotoredball; do timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w"; done | tee $OUT/bench_lines.txt
