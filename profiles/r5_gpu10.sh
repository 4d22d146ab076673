#!/bin/bash
# Round 5: the PACKED lane refill (k_refill_lane_packed: requests numbered across the segments, whole wavefronts of busy lanes) -- the GPU suite on the
# tree that makes it the default for the maze / MultiRoom / sentence levels, then A/B inside the same library (MG_LANE_PACKED=0: wavefront per episode)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5j; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))
except Exception as ex: print('$1 FAILED', ex)"; }
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 | tee $OUT/pytest_gpu_full_suite.log
for w in babyai_goto multiroom bosslevel; do
  for cfg in "MG_LANE_PACKED=0" "MG_LANE_LPW=64" "MG_LANE_LPW=32" "MG_LANE_LPW=16" "MG_LANE_LPW=8"; do
    env $cfg timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $cfg"
  done
done | tee $OUT/ab_packed_lane_refill.txt
for w in keycorridor unlockpickup gotoredball doorkey8x8 lavacrossing_full; do
  for cfg in "MG_LANE_PACKED=-1" "MG_LANE_PACKED=1" "MG_LANE_PACKED=1 MG_LANE_LPW=16"; do
    env $cfg timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $cfg"
  done
done | tee -a $OUT/ab_packed_lane_refill.txt
cd /tmp
for w in babyai_goto bosslevel multiroom; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $ROOT/bench.py --workload $w --steps 256 --warmup 64 --no-cpu-baseline > $OUT/prof_$w.log 2>&1
  f=$(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { cut -c1-170 $f | head -7 > $OUT/kernel_stats_$w.txt; cat $OUT/kernel_stats_$w.txt; }
  rm -rf $OUT/prof_$w
done
