#!/bin/bash
# Round 6, call 21: (1) where the dispatcher puts the waves of a k_roll7-shaped launch (profiles/microbench/simd_placement.hip); (2) the instruction records in
# the lane-contiguous layout (mg_device.h InstrAcc): GPU suite + BossLevel / other sentence levels
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/simd_placement.hip -o /tmp/simd_placement 2>/dev/null
{ /tmp/simd_placement 1024 36 256; /tmp/simd_placement 4096 36 256; /tmp/simd_placement 512 36 256; /tmp/simd_placement 1024 20 128; /tmp/simd_placement 1024 36 192; } > $OUT/simd_placement.txt 2>&1
head -40 $OUT/simd_placement.txt
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 > $OUT/pytest_gpu_call21.log; tail -3 $OUT/pytest_gpu_call21.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for w in bosslevel bosslevel babyai_goto; do
  timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w"
done | tee $OUT/bench_lines_instr_layout.txt
