#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_roll.py -q -k "pickl or wrapping or split" > $OUT/pytest_new.log 2>&1; echo "rc=$?"; tail -8 $OUT/pytest_new.log | cut -c1-200
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %-20s n=%-7d %7.3f G steps/s %6.2f us/step" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
for cfg in "4 0.12" "3 0.12" "3 0.2" "4 0.2" "4 0.3" "2 0.2"; do set -- $cfg
  for w in empty8x8 doorkey8x8 gotoredball; do
    MG_ROLL_NW=$1 MG_ROLL_RATIO=$2 timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json ${w}_nw$1_r$2
  done
done 2>&1 | tee $OUT/sweep_nw_ratio.txt
