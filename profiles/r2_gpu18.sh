#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2r; mkdir -p $O
export TMPDIR=/tmp
for w in empty8x8 doorkey8x8; do
  for lpe in 1 4; do
    MG_LPE=$lpe timeout 200 python bench.py --workload $w --fused 0 --steps 300 --warmup 60 --no-cpu-baseline > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$w unfused lpe=$lpe us/step %.2f' % (d['ms_per_step']*1e3))" | tee -a $O/unfused_lpe.txt
  done
done
timeout 2400 python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt; tail -6 $O/t_all.log | cut -c1-300
