#!/bin/bash
# round 2, GPU call 9: k_render_small, LPE=2, ring 32
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "rgb or fused_rollout_equals or spare_ring" > $O/t_sel.log 2>&1; echo "selected rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_sel.log
run() { name=$1; shift
  python bench.py "$@" --no-cpu-baseline > $O/b_$name.json 2> $O/b_$name.err
  python -c "
import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('$name', '%.3f G' % (d['value']/1e9), '%.2f us/step' % d['roofline']['avg_step_us'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/summary.txt
}
run rgb_small --workload empty8x8_rgb --steps 200 --warmup 20
MG_RENDER_SMALL=0 run rgb_old --workload empty8x8_rgb --steps 200 --warmup 20
run rgbp_small --workload doorkey8x8_rgb_partial --steps 200 --warmup 20
MG_RENDER_SMALL=0 run rgbp_old --workload doorkey8x8_rgb_partial --steps 200 --warmup 20
for lpe in 1 2; do MG_LPE=$lpe run empty_l$lpe --workload empty8x8 --steps 2048 --warmup 256; MG_LPE=$lpe run doorkey_l$lpe --workload doorkey8x8 --steps 2048 --warmup 256; done
run goto --workload gotoredball --steps 2048 --warmup 256
run lava --workload lavacrossing_full --steps 2048 --warmup 256
MG_LPE=2 run goto_l2 --workload gotoredball --steps 2048 --warmup 256
