#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2ac; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], '%.3f G steps/s %.2f us/step frac %.3f spl %s' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['steps_per_launch']))" $1 "$2" | tee -a $O/sweep_ring_sync.txt; }
for cfg in "64 8" "64 16" "128 8" "128 16"; do set -- $cfg
  MG_SPARE_RING=$1 MG_REFILL_WPS=$2 timeout 200 python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "gotoredball R=$1 wps=$2"
done
for cfg in "128 32" "128 64"; do set -- $cfg
  for w in doorkey8x8 lavacrossing_full; do
    MG_SPARE_RING=$1 MG_REFILL_WPS=$2 timeout 200 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "$w R=$1 wps=$2"
  done
done
