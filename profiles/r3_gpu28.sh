#!/bin/bash
# round 3: DynamicObstacles -- the in-place redraw beside the obstacle moves (two streams), envs per wavefront of k_move_obstacles
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3ab; mkdir -p $OUT
echo "== parity"
timeout 900 python -m pytest tests -m gpu -q -x -k "Dynamic or dynobs" 2>&1 | tail -3 | tee $OUT/pytest_dynobs.log
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-52s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run() { env "$@" timeout 100 python bench.py --workload dynobs16x16 --steps 256 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "dynobs16x16 $*"; }
echo "== dynobs16x16" | tee $OUT/dynobs_overlap.txt
{ run MG_LIVE_OVERLAP=0; run MG_LIVE_OVERLAP=1; run MG_LIVE_OVERLAP=1 MG_MOVE_EPB=8; run MG_LIVE_OVERLAP=1 MG_MOVE_EPB=32; run MG_LIVE_OVERLAP=1 MG_MOVE_EPB=64; run MG_LIVE_OVERLAP=0 MG_MOVE_EPB=32; run MG_LIVE_OVERLAP=0 MG_MOVE_EPB=64; } 2>&1 | tee -a $OUT/dynobs_overlap.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o x -- python $ROOT/bench.py --workload dynobs16x16 --steps 256 --warmup 64 --no-cpu-baseline > $OUT/prof.log 2>&1
head -6 $(find $OUT/prof -name '*kernel_stats.csv' | head -1) | cut -c1-150 | tee $OUT/kernel_stats_dynobs16x16.csv; rm -rf $OUT/prof
