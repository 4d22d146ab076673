#!/bin/bash
# Round 5, GPU call 4: the tree with (a) the Unlock family + KeyCorridor on the lane generators, (b) the ONE-step specialisation of k_roll7:
# the whole GPU suite, then what each buys (A/B inside the SAME library: MG_LANE_GEN=0 / MG_ROLL_ONE=0 select the previous paths).
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5d; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))
except Exception as ex: print('$1 FAILED', ex)"; }
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 | tee $OUT/pytest_gpu_full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
for rep in 1 2; do for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do for one in 1 0; do
  MG_ROLL_ONE=$one timeout 200 python bench.py --workload $w --fused 0 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | line "$w one-launch-per-step MG_ROLL_ONE=$one"
done; done; done | tee $OUT/ab_one_step.txt
for w in keycorridor unlock unlockpickup blockedunlockpickup; do for lg in 1 0; do
  MG_LANE_GEN=$lg timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w MG_LANE_GEN=$lg"
done; done | tee $OUT/ab_lane_product.txt
cd /tmp
for one in 1 0; do
  MG_ROLL_ONE=$one timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_one$one -o e -- python $ROOT/bench.py --fused 0 --steps 500 --warmup 50 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $OUT/prof_one$one -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { echo "MG_ROLL_ONE=$one"; grep k_roll7 $f | cut -c1-200; } | tee -a $OUT/kernel_stats_one_step.txt
  rm -rf $OUT/prof_one$one
done
cd $ROOT
