#!/bin/bash
# Round 6, call 31: the FullyObs encode in blocks of four rounds (code dwords, then lookups, then packs + stores): parity of the FullyObs tests, then
# LavaCrossing FullyObs x 131 072 with two / three / four waves per workgroup, and other FullyObs workloads
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_full.py tests/test_gpu_roll.py tests/test_gpu_launch_lengths.py -q -m gpu -n 4 2>&1 | tail -3 | tee $OUT/pytest_gpu_call31.log
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %s' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0')))
except Exception as ex: print('$1 FAILED', ex)"; }
for v in "MG_X=0" "MG_ROLL_NW=3" "MG_ROLL_NW=4" "MG_X=0" "MG_ROLL_NW=3"; do
  env $v python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full $v"
done | tee $OUT/ab_fullyobs_encode_blocks.txt
for w in empty8x8 doorkey8x8 keycorridor; do
  python bench.py --workload $w --obs-mode full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w FullyObs"
done | tee -a $OUT/ab_fullyobs_encode_blocks.txt
python bench.py --workload lavacrossing_full --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full one launch per step" | tee -a $OUT/ab_fullyobs_encode_blocks.txt
