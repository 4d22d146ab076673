#!/bin/bash
# round 3: bench.py after the kernel-name change -- every workload prints its line
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3ae; mkdir -p $OUT
for w in empty8x8 lavacrossing_full dynobs16x16 empty8x8_rgb bosslevel; do
  timeout 120 python bench.py --workload $w --steps 40 --warmup 8 --no-cpu-baseline > $OUT/b_$w.json 2> $OUT/b_$w.err || { echo "$w FAILED"; tail -5 $OUT/b_$w.err; }
  python -c "
import json; d=json.loads(open('$OUT/b_$w.json').read().strip().splitlines()[-1]); print('$w', d['roofline']['kernel'], '|', d['config']['launch'][:60], '| %.2f G' % (d['value']/1e9))"
done
timeout 100 python bench.py --workload empty8x8 --view 5 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('view5', d['roofline']['kernel'], '| %.2f G' % (d['value']/1e9))"
timeout 100 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -c 300
