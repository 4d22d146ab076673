#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "GoTo-v0 or GoToOpen or GoToObjMaze or BabyAI-Pickup-v0 or BabyAI-Open-v0" > $O/t_maze.log 2>&1; echo "maze rc=$?" | tee -a $O/summary.txt; tail -15 $O/t_maze.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt; tail -8 $O/t_all.log | cut -c1-300
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver$i.json 2> $O/bench_driver.err; python -c "
import json; d=json.loads(open('$O/bench_driver$i.json').read().strip().splitlines()[-1]); print('driver-like', d['value']/1e9, d['ms_per_step'])" | tee -a $O/summary.txt; done
