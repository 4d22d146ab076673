#!/bin/bash
# ONE command for a node with more than one MI355X (the round's gpurun boxes have one; the driver's 8-GPU node runs this first):
# the nccl (= RCCL over xGMI) tests of the sharded env, then bench.py at 2 / 4 / 8 GPUs with and without the optional per-launch
# all-gather of the step records.  bench.py spawns its ranks itself (one process per GPU, 127.0.0.1 rendezvous).
#   bash profiles/run_multi.sh [max_gpus] > multi.log
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
MAXG=${1:-$(python -c "import torch; print(torch.cuda.device_count())")}
OUT=gpurun_out/multi; mkdir -p $OUT
echo "== $MAXG GPUs visible"
timeout 1800 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_gpu_multi.log
for g in 1 2 4 8; do
  [ "$g" -le "$MAXG" ] || continue
  for gather in 0 1; do
    [ "$g" -eq 1 ] && [ "$gather" -eq 1 ] && continue
    for w in empty8x8 lavacrossing_full gotoredball; do
      timeout 600 python bench.py --gpus $g --workload $w --steps 1024 --warmup 128 --gather-obs $gather > $OUT/bench_${w}_g${g}_gather${gather}.json 2> $OUT/bench_${w}_g${g}_gather${gather}.err
      python - $OUT/bench_${w}_g${g}_gather${gather}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-60s %8.3f G env-steps/s  %7.2f us/step  world %s backend %s" % (sys.argv[1], d["value"] / 1e9, d["ms_per_step"] * 1e3,
          d["config"]["distributed"]["world_size"], d["config"]["distributed"]["backend"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
    done
  done
done | tee $OUT/summary.txt
