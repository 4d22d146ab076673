#!/bin/bash
# DynamicObstacles step with the LDS-staged k_move_obstacles: tests, bench, per-kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "Dynamic or dynobs or DynObs or dynamic" > $O/t_dyn.log 2>&1; echo "dyn tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_dyn.log
timeout 200 python bench.py --workload dynobs16x16 --steps 300 --warmup 60 --no-cpu-baseline > $O/bench_dynobs_after.json 2> $O/bench_dynobs_after.err
python -c "
import json; d=json.loads(open('$O/bench_dynobs_after.json').read().strip().splitlines()[-1]); print('dynobs16x16 after: %.3f G steps/s, %.2f us/step' % (d['value']/1e9, d['ms_per_step']*1e3))" | tee -a $O/summary.txt
cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_after -o dyn -- python $GRAFT_REPO_ROOT/bench.py --workload dynobs16x16 --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find /tmp/prof_after -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_dynobs_after.csv 2>/dev/null; head -8 $O/kernel_stats_dynobs_after.csv | cut -c1-170
