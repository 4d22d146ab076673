#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2x; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac']))" $1 "$2" | tee -a $O/summary.txt; }
for w in gotoredball empty8x8 doorkey8x8 lavacrossing_full; do timeout 200 python bench.py --workload $w --steps 2000 --warmup 300 --no-cpu-baseline > $O/bench_$w.json 2> $O/b.err; show $O/bench_$w.json $w; done
timeout 2400 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt; tail -4 $O/t_all.log | cut -c1-300
