#!/bin/bash
# round 2, GPU call 5: whole GPU suite (ObstructedMaze ids included), then the round's evidence (bench + kernel stats + PMC)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_all.log
bash profiles/collect.sh r2e empty8x8 doorkey8x8 lavacrossing_full gotoredball 2>&1 | tee -a $O/summary.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver$i.json 2> $O/bench_driver.err; python -c "
import json; d=json.loads(open('$O/bench_driver$i.json').read().strip().splitlines()[-1]); print('driver-like', d['value']/1e9, d['ms_per_step'])" | tee -a $O/summary.txt; done
for w in empty8x8 doorkey8x8 gotoredball; do timeout 200 python bench.py --workload $w --fused 0 --steps 2000 --warmup 300 --no-cpu-baseline > $O/bench_${w}_unfused.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_${w}_unfused.json').read().strip().splitlines()[-1]); print('$w unfused', d['value']/1e9, d['roofline']['avg_step_us'])" | tee -a $O/summary.txt; done
