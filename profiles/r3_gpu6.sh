#!/bin/bash
# round 3, GPU call 6: full GPU suite on the shuffle fix + in-place wrappers + async ring fill; slow-family bench lines; guard loop.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT
echo "== first process of the box, red zones on"
MG_GUARD=1 timeout 180 python profiles/first_process.py > $OUT/first_process_guard.log 2>&1; echo "rc=$? $(tail -1 $OUT/first_process_guard.log)"
echo "== full GPU suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee $OUT/pytest_full.log
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %-20s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== bench lines" | tee $OUT/bench.txt
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do timeout 120 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err; line $OUT/bench_$w.json $w; done 2>&1 | tee -a $OUT/bench.txt
for w in dynobs16x16 bosslevel; do timeout 150 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err; line $OUT/bench_$w.json $w; done 2>&1 | tee -a $OUT/bench.txt
timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/b.err; line $OUT/bench_driver.json driver_sized | tee -a $OUT/bench.txt
timeout 60 python bench.py --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/bench_unfused.json 2> $OUT/b.err; line $OUT/bench_unfused.json empty_unfused | tee -a $OUT/bench.txt
python - <<'PY' 2>&1 | tee $ROOT/gpurun_out/r3f/reset_latency.txt
import time, sys
sys.path.insert(0, ".")
import minigrid_amd as mg
env = mg.make_vec("MiniGrid-DoorKey-8x8-v0", 262144, output="torch")
env.reset(seed=0); env.sync()
for k in range(3):
    t0 = time.perf_counter(); obs, _ = env.reset(seed=k + 1); import torch; torch.cuda.synchronize(); t1 = time.perf_counter()
    env.sync(); t2 = time.perf_counter()
    print(f"DoorKey-8x8 x 262144 reset(seed): first observation after {1e3*(t1-t0):.2f} ms, ring redrawn after {1e3*(t2-t0):.2f} ms")
env.close()
PY
echo "== crash hunt: fresh processes with red zones (MG_GUARD=1)"
i=0; bad=0; t_end=$((SECONDS + 200))
while [ $SECONDS -lt $t_end ] && [ $i -lt 40 ]; do
  i=$((i+1)); MG_GUARD=1 timeout 120 python profiles/first_process.py 32768 > $OUT/guard_run.log 2>&1; rc=$?
  echo "guard run $i rc=$rc $(tail -1 $OUT/guard_run.log | cut -c1-60)"
  if [ $rc -ne 0 ]; then bad=$((bad+1)); cp $OUT/guard_run.log $OUT/guard_run_FAILED_$i.log; fi
done 2>&1 | tee $OUT/guard_loop.txt
