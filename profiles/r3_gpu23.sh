#!/bin/bash
# round 3: the short-episode FullyObs shard (LavaCrossing): launch length, staged spares, waves per workgroup
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3w; mkdir -p $OUT
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-52s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f spl %d" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"], d["config"]["steps_per_launch"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== lavacrossing_full: launch length / shadows / waves" | tee $OUT/lava_sweep.txt
run() { env "$@" timeout 100 python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "lavacrossing_full $*"; }
{ run X=0; run MG_TRAJ_SLOTS=16; run MG_TRAJ_SLOTS=8; run MG_ROLL_SHADOWS=2; run MG_ROLL_SHADOWS=2 MG_TRAJ_SLOTS=16; run MG_ROLL_NW=1; run MG_ROLL_NW=3; run MG_ROLL_NW=4; run MG_NO_ROLL_FULL=1; } 2>&1 | tee -a $OUT/lava_sweep.txt
echo "== lavacrossing partial obs (k_roll7<0,false>)" | tee -a $OUT/lava_sweep.txt
for e in X=0 MG_TRAJ_SLOTS=16 MG_ROLL_SHADOWS=2; do env $e timeout 100 python bench.py --workload lavacrossing_full --obs-mode partial --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "lavacrossing partial $e"; done 2>&1 | tee -a $OUT/lava_sweep.txt
