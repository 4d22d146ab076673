#!/bin/bash
# round 4: evidence of the final build -- the GPU suite, smoke(), the profiler passes (kernel trace + separate FETCH_SIZE / WRITE_SIZE
# passes, 32-step launches for the four BASELINE workloads and the driver's 20-step launch shape for the headline), the bench lines that
# quote them, the other workloads, SQ counters of the headline
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4final; mkdir -p $OUT $ROOT/profiles/r4
python -c "from minigrid_amd import build; print('library stale:', build._stale())" | tee $OUT/build_state.txt
# the crash-hunt reproducer as the FIRST GPU process of this fresh box (every round-2 abort hit a box's first process)
MG_GUARD=1 timeout 300 python profiles/first_process.py 65536 > $OUT/first_process_guarded.log 2>&1; echo "first_process rc=$?" | tee -a $OUT/first_process_guarded.log | tail -1
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > $OUT/pytest_gpu_full_suite.log; tail -3 $OUT/pytest_gpu_full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
bash profiles/collect_r4.sh r4final "long" empty8x8 doorkey8x8 lavacrossing_full gotoredball dynobs16x16 2>&1 | grep -E "^\{|SIZE" | cut -c1-400
bash profiles/collect_r4.sh r4final "spl20" empty8x8 2>&1 | grep -E "^\{|SIZE" | cut -c1-400
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.2f us/step (event %.2f)  frac %.3f  traffic %s  host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['traffic'], (d['host_ms']-d['event_ms'])*1e3))"; }
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  extra="--no-cpu-baseline"; [ $w = empty8x8 ] && extra=""
  timeout 300 python bench.py --workload $w --steps 2048 --warmup 256 $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  cat $OUT/bench_$w.json | line "$w"
done | tee $OUT/bench_lines_baseline_workloads.txt
for i in 1 2 3; do timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver$i.json 2> $OUT/bench_driver$i.err; cat $OUT/bench_driver$i.json | line "driver-sized (--steps 20 --warmup 5)"; done | tee $OUT/bench_lines_driver.txt
timeout 300 python bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err; cat $OUT/bench_default_run.json | line "default run (no flags)" | tee -a $OUT/bench_lines_driver.txt
for w in bosslevel dynobs16x16 dynobs8x8 dynobs6x6 empty8x8_rgb doorkey8x8_rgb_partial; do
  timeout 300 python bench.py --workload $w --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_$w.json | line "$w"
done | tee $OUT/bench_lines_other_workloads.txt
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball dynobs16x16 bosslevel; do
  timeout 300 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_${w}_unfused.json | line "$w one launch per step"
done | tee -a $OUT/bench_lines_other_workloads.txt
bash profiles/pmc_sq_r4.sh r4final empty8x8 final > /dev/null 2>&1
cp $OUT/sq_counters_empty8x8_final.txt $OUT/sq_counters_empty8x8.txt 2>/dev/null
ls $OUT | head -80
