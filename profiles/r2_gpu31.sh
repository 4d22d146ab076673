#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2af; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac']))" $1 "$2" | tee -a $O/summary.txt; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "Empty-8x8 or DoorKey-8x8 or GoToRedBall-v0 or ragged or full_size_config2" > $O/t_sel.log 2>&1; echo "selected tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_sel.log | cut -c1-300
for w in empty8x8 doorkey8x8 gotoredball; do timeout 200 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline > $O/bench_$w.json 2> $O/b.err; show $O/bench_$w.json $w; done
timeout 200 python bench.py --workload empty8x8 --fused 0 --steps 400 --warmup 100 --no-cpu-baseline > $O/bench_empty_unfused.json 2> $O/b.err; show $O/bench_empty_unfused.json "empty8x8 unfused"
