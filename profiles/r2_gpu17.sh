#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
for w in empty8x8_rgb doorkey8x8_rgb_partial; do
  for thr in 0 1 2 4; do
    for blocks in default 2048; do
      if [ $blocks = default ]; then unset MG_RENDER_BLOCKS; else export MG_RENDER_BLOCKS=$blocks; fi
      MG_RENDER_THROTTLE=$thr timeout 200 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline > $O/b.json 2> $O/b.err
      python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$w throttle=$thr blocks=$blocks us/step %.1f frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" | tee -a $O/render_throttle.txt
    done
  done
done
