#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3m; mkdir -p $OUT
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %-20s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== quick: SAME_STEP, shadows, fused"
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_quick.log
echo "== one vs two staged spares" | tee $OUT/shadows.txt
for sh in 1 2; do
  for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do MG_ROLL_SHADOWS=$sh timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json ${w}_shadows$sh; done
  MG_ROLL_SHADOWS=$sh timeout 100 python bench.py --workload lavacrossing_full --obs-mode partial --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_partial_shadows$sh
done 2>&1 | tee -a $OUT/shadows.txt
for nw in 2 3; do MG_ROLL_NW=$nw timeout 100 python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json lava_full_nw$nw; done 2>&1 | tee -a $OUT/shadows.txt
echo "== full GPU suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_full.log
