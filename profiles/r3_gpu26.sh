#!/bin/bash
# round 3, final build: the crash-hunt script as the FIRST GPU process of a fresh box, then the multi-process paths of bench.py with
# two ranks sharing the one GPU (gloo): no collective, per-step gather, fused block gather; the nccl / multi-GPU tests (skipped here)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3y; mkdir -p $OUT
echo "== first process of this box" | tee $OUT/first_process_final.txt
timeout 300 python profiles/first_process.py 2>&1 | tail -6 | tee -a $OUT/first_process_final.txt
MG_GUARD=1 timeout 300 python profiles/first_process.py 2>&1 | tail -3 | tee -a $OUT/first_process_final.txt
echo "== two ranks on one GPU (gloo)" | tee $OUT/bench_2rank_gloo.txt
run2() { name=$1; shift; (cd /tmp && timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT $ROOT/bench.py --gpus 2 --backend gloo --no-cpu-baseline "$@" > $OUT/bench_2rank_$name.json 2> $OUT/bench_2rank_$name.err); python - $OUT/bench_2rank_$name.json "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s n_gpus(ranks) %d  %7.3f G steps/s %6.2f us/step  per-rank us %s  %s" % (sys.argv[2], d["n_gpus"], d["value"]/1e9, d["ms_per_step"]*1e3, [round(x,2) for x in d.get("per_rank_us_per_step", [])], d.get("distributed")))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
{ PORT=29541 run2 no_collective --steps 2048 --warmup 256
  PORT=29542 run2 fused_block_gather --steps 1024 --warmup 128 --gather-obs 1
  PORT=29543 run2 per_step_gather --steps 256 --warmup 32 --gather-obs 1 --fused 0
  PORT=29544 run2 driver_sized --steps 20 --warmup 5; } 2>&1 | tee -a $OUT/bench_2rank_gloo.txt
echo "== multi-GPU tests (need >= 2 GPUs: skipped on this box) and the single-process stream test"
timeout 300 python -m pytest tests/test_gpu_multi.py -m gpu -q -rs 2>&1 | tail -6 | tee $OUT/pytest_multi.log
