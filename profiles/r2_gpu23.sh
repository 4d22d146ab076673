#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "Dynamic or dynobs or DynObs or dynamic" > $O/t_dyn.log 2>&1; echo "dyn tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/t_dyn.log
timeout 200 python bench.py --workload dynobs16x16 --steps 300 --warmup 60 --no-cpu-baseline > $O/bench_dynobs16x16.json 2> $O/b.err
python -c "
import json; d=json.loads(open('$O/bench_dynobs16x16.json').read().strip().splitlines()[-1]); print('dynobs16x16: %.3f G steps/s, %.2f us/step' % (d['value']/1e9, d['ms_per_step']*1e3))" | tee -a $O/summary.txt
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dyn -o dyn -- python $GRAFT_REPO_ROOT/bench.py --workload dynobs16x16 --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; cp $(find /tmp/prof_dyn -name '*kernel_stats.csv' | head -1) $O/kernel_stats_dynobs16x16.csv; head -5 $O/kernel_stats_dynobs16x16.csv | cut -c1-150
