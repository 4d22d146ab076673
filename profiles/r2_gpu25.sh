#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ab; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f spl %s' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['steps_per_launch']))" $1 "$2" | tee -a $O/sweep_ring_defaults.txt; }
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_philox.py -m gpu -q -x > $O/t_fused.log 2>&1; echo "fused+philox rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_fused.log | cut -c1-300
for w in empty8x8 gotoredball doorkey8x8 lavacrossing_full; do
  timeout 200 python bench.py --workload $w --steps 1536 --warmup 256 --no-cpu-baseline > $O/bench_$w.json 2> $O/b.err; show $O/bench_$w.json "$w defaults (R=64 S=32)"
  if [ $w != empty8x8 ]; then MG_SPARE_RING=128 MG_TRAJ_SLOTS=64 timeout 200 python bench.py --workload $w --steps 1536 --warmup 256 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "$w R=128 S=64"; fi
done
for i in 1 2; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver$i.json 2> $O/b.err; show $O/bench_driver$i.json "driver-sized"; done
timeout 2400 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt; tail -4 $O/t_all.log | cut -c1-300
