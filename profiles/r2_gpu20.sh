#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2t; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k "KeyInBox" > $O/t_kib.log 2>&1; echo "keyinbox rc=$?" | tee -a $O/summary.txt; tail -5 $O/t_kib.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt; tail -8 $O/t_all.log | cut -c1-300
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver$i.json 2> $O/bench_driver.err; python -c "
import json; d=json.loads(open('$O/bench_driver$i.json').read().strip().splitlines()[-1]); print('driver-like', d['value']/1e9, d['ms_per_step'])" | tee -a $O/summary.txt; done
