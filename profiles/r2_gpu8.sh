#!/bin/bash
# round 2, GPU call 8: refill wps, LPE by batch size, k_render geometry sweep (short-lived workgroups), 2-rank gloo bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
run() { # name, env..., -- args
  name=$1; shift
  python bench.py "$@" --no-cpu-baseline > $O/b_$name.json 2> $O/b_$name.err
  python -c "
import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('$name', '%.3f G' % (d['value']/1e9), '%.2f us/step' % d['roofline']['avg_step_us'], 'frac %.3f' % d['roofline']['frac'])" | tee -a $O/summary.txt
}
for lpe in 4 1; do for wps in 4 16; do MG_LPE=$lpe MG_REFILL_WPS=$wps run goto_l${lpe}_w$wps --workload gotoredball --steps 2048 --warmup 256; done; done
run goto_default --workload gotoredball --steps 2048 --warmup 256
run lava_default --workload lavacrossing_full --steps 2048 --warmup 256
MG_SPARE_RING=32 run goto_ring32 --workload gotoredball --steps 2048 --warmup 256
run rgb_default --workload empty8x8_rgb --steps 200 --warmup 20
for epw in 1 2 4 8 16; do MG_RENDER_EPW=$epw MG_RENDER_BLOCKS=10000000 run rgb_short_epw$epw --workload empty8x8_rgb --steps 200 --warmup 20; done
for blocks in 512 1024 2048; do MG_RENDER_EPW=16 MG_RENDER_BLOCKS=$blocks run rgb_persist_b$blocks --workload empty8x8_rgb --steps 200 --warmup 20; done
cd /tmp && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 200 --warmup 20 --backend gloo --gather-obs 1 > $GRAFT_REPO_ROOT/$O/b_2rank_gloo.json 2> $GRAFT_REPO_ROOT/$O/b_2rank_gloo.err; tail -c 600 $GRAFT_REPO_ROOT/$O/b_2rank_gloo.json; tail -3 $GRAFT_REPO_ROOT/$O/b_2rank_gloo.err
