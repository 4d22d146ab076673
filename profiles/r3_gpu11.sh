#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3k; mkdir -p $OUT
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %-20s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== FullyObs through k_roll7: quick parity"
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_full.py -x -q -k "True or full or Lava or wrapping or pickl" 2>&1 | tail -6 | tee $OUT/pytest_full_quick.log
echo "== FullyObs bench" | tee $OUT/full_bench.txt
MG_NO_ROLL_FULL=1 timeout 100 python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_k_step_lpe4 | tee -a $OUT/full_bench.txt
for nw in 1 2 3 4; do MG_ROLL_NW=$nw timeout 100 python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_roll_nw$nw; done 2>&1 | tee -a $OUT/full_bench.txt
for x in 2 16; do MG_ROLL_NW=3 MG_EXP=$x timeout 100 python bench.py --workload lavacrossing_full --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_nw3_exp$x; done 2>&1 | tee -a $OUT/full_bench.txt
timeout 100 python bench.py --workload lavacrossing_full --fused 0 --steps 256 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_unfused | tee -a $OUT/full_bench.txt
timeout 100 python bench.py --workload doorkey8x8 --obs-mode full --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json doorkey_full | tee -a $OUT/full_bench.txt
echo "== driver-sized and one-launch-per-step after the share policy / event warm-up"
for i in 1 2 3; do timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json driver_sized_$i; done 2>&1 | tee $OUT/driver.txt
for w in empty8x8 doorkey8x8; do timeout 100 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json ${w}_unfused; done 2>&1 | tee -a $OUT/driver.txt
echo "== full GPU suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_full.log
