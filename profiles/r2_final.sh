#!/bin/bash
# Round-2 evidence on the final build: the four BASELINE workloads (bench + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes),
# driver-sized runs, the single-step path, the RGB workloads, DynamicObstacles with its per-kernel breakdown.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash profiles/collect.sh r2ag empty8x8 doorkey8x8 lavacrossing_full gotoredball
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2ag
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac']))" $1 "$2" | tee -a $O/summary.txt; }
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver$i.json 2> $O/bench_driver.err; show $O/bench_driver$i.json "driver-sized"; done
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do timeout 200 python bench.py --workload $w --fused 0 --steps 400 --warmup 100 --no-cpu-baseline > $O/bench_${w}_unfused.json 2> $O/b.err; show $O/bench_${w}_unfused.json "$w unfused"; done
for w in empty8x8_rgb doorkey8x8_rgb_partial dynobs16x16; do timeout 200 python bench.py --workload $w --steps 300 --warmup 60 --no-cpu-baseline > $O/bench_$w.json 2> $O/b.err; show $O/bench_$w.json "$w"; done
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dyn -o dyn -- python $GRAFT_REPO_ROOT/bench.py --workload dynobs16x16 --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; cp $(find /tmp/prof_dyn -name '*kernel_stats.csv' | head -1) $O/kernel_stats_dynobs16x16.csv; head -7 $O/kernel_stats_dynobs16x16.csv | cut -c1-170
