#!/bin/bash
# round 2, GPU call 6: PutNear (16-bit mission ids) + everything touching missions, refill stream priority / 8 waves per segment
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -k "PutNear or fused or philox or sharded or dict_observation or torch_views or GoToObject or Fetch-5x5 or stepping_past" > $O/t_sel.log 2>&1; echo "selected rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_sel.log
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  timeout 300 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', '%.3f G' % (d['value']/1e9), '%.2f us/step' % d['roofline']['avg_step_us'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/summary.txt
done
for w in lavacrossing_full gotoredball; do bash profiles/kstats.sh $w 2>&1 | tee $O/kstats_$w.txt | tail -9; done
