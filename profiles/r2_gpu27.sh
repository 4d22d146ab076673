#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ac; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f spl %s' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['steps_per_launch']))" $1 "$2" | tee -a $O/sweep_ring_sync.txt; }
timeout 600 python -m pytest tests/test_gpu_fused.py -m gpu -q -x > $O/t_fused.log 2>&1; echo "fused rc=$?" | tee -a $O/summary.txt; tail -2 $O/t_fused.log | cut -c1-200
for i in 1 2 3; do
  for R in 64 128; do
    MG_SPARE_RING=$R timeout 200 python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "gotoredball R=$R run $i"
  done
done
for w in doorkey8x8 lavacrossing_full; do
  for R in 64 128; do
    MG_SPARE_RING=$R timeout 200 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "$w R=$R"
  done
done
