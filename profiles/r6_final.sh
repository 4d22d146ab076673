#!/bin/bash
# Round 6: evidence of the final build.  The profiler passes (kernel trace + separate FETCH_SIZE / WRITE_SIZE passes; 32-step launches and the driver's
# 20-step launch shape over twenty launches, both for the four BASELINE workloads) carry step_kernel_srchash: bench.py quotes them only on this build
# of the step kernels.  Then the bench lines that quote them, the driver-shaped line, one launch per step, the other workloads.
#   usage: bash profiles/r6_final.sh [skip-suite]
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6final; mkdir -p $OUT $ROOT/profiles/r6
python -c "
from minigrid_amd import build; print('library stale:', build._stale(), ' step_kernel_srchash:', build.step_kernel_hash())" | tee $OUT/build_state.txt
if [ "$1" != "skip-suite" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > $OUT/pytest_gpu_full_suite.log; tail -3 $OUT/pytest_gpu_full_suite.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
fi
bash profiles/collect_r6.sh r6final "long spl20" empty8x8 doorkey8x8 lavacrossing_full gotoredball 2>&1 | grep -E "^\{|SIZE" | cut -c1-300
bash profiles/collect_r6.sh r6final "long" dynobs16x16 keycorridor babyai_goto multiroom bosslevel 2>&1 | grep -E "^\{|SIZE" | cut -c1-300
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f)  frac this run %.3f / profile %s  valu %s  traffic %s (%s)  host-event %.1f us  episodes in the timed region %s (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], ('%.3f' % r['frac_profile']) if r.get('frac_profile') else None, ('%.3f' % r['valu_issue_frac']) if r.get('valu_issue_frac') else None, r['traffic'], (r['traffic_source'] or 'floor'), (d['host_ms']-d['event_ms'])*1e3, c.get('episodes_finished_in_timed_region_rank0'), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  extra="--no-cpu-baseline"; [ $w = empty8x8 ] && extra=""
  timeout 300 python bench.py --workload $w --steps 2048 --warmup 256 $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  cat $OUT/bench_$w.json | line "$w"
done | tee $OUT/bench_lines_baseline_workloads.txt
for i in 1 2 3; do timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver$i.json 2> $OUT/bench_driver$i.err; cat $OUT/bench_driver$i.json | line "driver-sized (--steps 20 --warmup 5)"; done | tee $OUT/bench_lines_driver.txt
for w in doorkey8x8 lavacrossing_full gotoredball; do timeout 100 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_driver_$w.json | line "$w driver-sized"; done | tee -a $OUT/bench_lines_driver.txt
timeout 300 python bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err; cat $OUT/bench_default_run.json | line "default run (no flags)" | tee -a $OUT/bench_lines_driver.txt
for w in keycorridor unlock unlockpickup blockedunlockpickup multiroom babyai_goto bosslevel dynobs16x16 dynobs8x8 dynobs6x6 empty8x8_rgb doorkey8x8_rgb_partial; do
  timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_$w.json | line "$w"
done | tee $OUT/bench_lines_other_workloads.txt
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball dynobs16x16 bosslevel; do
  timeout 300 python bench.py --workload $w --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_${w}_unfused.json | line "$w one launch per step"
done | tee -a $OUT/bench_lines_other_workloads.txt
ls $OUT | wc -l
