#!/bin/bash
# round 2, GPU call 7: single-wave refill workgroups; store-stream micro-benchmark for k_render
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -q -x > $O/t_fused.log 2>&1; echo "fused rc=$?" | tee -a $O/summary.txt
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  timeout 300 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', '%.3f G' % (d['value']/1e9), '%.2f us/step' % d['roofline']['avg_step_us'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/summary.txt
done
for w in lavacrossing_full gotoredball doorkey8x8; do bash profiles/kstats.sh $w 2>&1 | tee $O/kstats_$w.txt | tail -9; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/storebench.hip -o /tmp/storebench && timeout 300 /tmp/storebench 2>&1 | tee $O/storebench.txt
