#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ae; mkdir -p $O
export TMPDIR=/tmp
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  extra="--no-cpu-baseline"; [ $w = empty8x8 ] && extra=""
  timeout 300 python bench.py --workload $w --steps 2000 --warmup 300 $extra > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json,sys; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$w %.3f G %.2f us/step frac %.3f traffic %.0f MB algo %.0f MB' % (d['value']/1e9, d['ms_per_step']*1e3, r['frac'], r['traffic']/1e6, r['algorithmic_bytes_per_launch']/1e6))"
done
timeout 200 python bench.py > $O/bench_default.json 2> $O/b.err; tail -c 600 $O/bench_default.json
