#!/bin/bash
# round 3: time-split parameters again, now that the own step is cheaper (quad encode): silent/own ratio, waves per workgroup
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3x; mkdir -p $OUT
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-52s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run() { w=$1; shift; env "$@" timeout 100 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$w $*"; }
echo "== split ratio / waves (quad encode)" | tee $OUT/sweep_nw_ratio_quads.txt
{ for r in 0.06 0.12 0.18 0.25 0.35; do run empty8x8 MG_ROLL_RATIO=$r; done
  for nw in 2 3 4; do for r in 0.12 0.25; do run doorkey8x8 MG_ROLL_NW=$nw MG_ROLL_RATIO=$r; done; done
  for nw in 2 3 4; do run gotoredball MG_ROLL_NW=$nw MG_ROLL_RATIO=0.25; done
  run empty8x8 MG_ROLL_NW=3 MG_ROLL_RATIO=0.12
  run empty8x8 X=0; run doorkey8x8 X=0; } 2>&1 | tee -a $OUT/sweep_nw_ratio_quads.txt
