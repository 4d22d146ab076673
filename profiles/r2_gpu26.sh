#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ab; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f spl %s' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['steps_per_launch']))" $1 "$2" | tee -a $O/sweep_ring_defaults.txt; }
for w in gotoredball doorkey8x8 lavacrossing_full; do
  for cfg in "128 32" "256 32" "256 64"; do
    set -- $cfg
    MG_SPARE_RING=$1 MG_TRAJ_SLOTS=$2 timeout 200 python bench.py --workload $w --steps 1536 --warmup 256 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "$w R=$1 S=$2"
  done
done
