#!/bin/bash
# round 2 evidence: bench + kernel stats + PMC for the four BASELINE workloads (profiles/collect.sh), driver-like run, unfused runs,
# RGB, SQ counters of the headline kernel, 2-rank gloo run of bench.py on one GPU; the fused tests twice more
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > $O/t_fused$i.log 2>&1; echo "fused run $i rc=$?" | tee -a $O/summary.txt; done
bash profiles/collect.sh r2k empty8x8 doorkey8x8 lavacrossing_full gotoredball 2>&1 | tee -a $O/summary.txt
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver$i.json 2> $O/bench_driver.err; python -c "
import json; d=json.loads(open('$O/bench_driver$i.json').read().strip().splitlines()[-1]); print('driver-like', d['value']/1e9, d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_python']['one_core'])" | tee -a $O/summary.txt; done
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do timeout 200 python bench.py --workload $w --fused 0 --steps 2000 --warmup 300 --no-cpu-baseline > $O/bench_${w}_unfused.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_${w}_unfused.json').read().strip().splitlines()[-1]); print('$w unfused', d['value']/1e9, d['roofline']['avg_step_us'])" | tee -a $O/summary.txt; done
for w in empty8x8_rgb doorkey8x8_rgb_partial; do timeout 300 python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value']/1e9, d['roofline']['avg_step_us'], d['roofline']['frac'])" | tee -a $O/summary.txt; done
bash profiles/pmc_sq.sh r2k empty8x8 2>&1 | tee $O/sq_counters_empty8x8.txt | tail -3
(cd /tmp && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 200 --warmup 20 --backend gloo --gather-obs 1 > $GRAFT_REPO_ROOT/$O/bench_2rank_gloo_gather.json 2> $GRAFT_REPO_ROOT/$O/bench_2rank_gloo.err; tail -c 700 $GRAFT_REPO_ROOT/$O/bench_2rank_gloo_gather.json)
(cd /tmp && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 2048 --warmup 256 --backend gloo > $GRAFT_REPO_ROOT/$O/bench_2rank_gloo.json 2>> $GRAFT_REPO_ROOT/$O/bench_2rank_gloo.err; tail -c 300 $GRAFT_REPO_ROOT/$O/bench_2rank_gloo.json)
