#!/bin/bash
# Round 5: the see-through fast path of obs7_view (one byte transpose instead of two for the levels that skip process_vis) -- what it buys
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5g; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
for rep in 1 2 3; do
  python bench.py --workload empty8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty8x8"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "empty8x8 driver-shape"
done | tee $OUT/see_through.txt
for w in doorkey8x8 dynobs16x16 dynobs8x8 dynobs6x6 gotoredball; do python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w"; done | tee -a $OUT/see_through.txt
python bench.py --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | line "empty8x8 one launch per step" | tee -a $OUT/see_through.txt
timeout 600 python -m pytest tests/test_gpu_roll.py tests/test_gpu_dynobs.py tests/test_gpu_lds_protocol.py -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_subset.log
