#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3n; mkdir -p $OUT
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-30s %-20s n=%-7d %7.3f G steps/s %6.2f us/step spl %d" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["config"]["steps_per_launch"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== sentence levels with the verifier in the step loop"
timeout 900 python -m pytest tests/test_gpu_roll.py -x -q -k "sentence" 2>&1 | tail -8 | tee $OUT/pytest_sentence_fused.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_philox.py -x -q -k "BossLevel or GoToSeq or Synth or OpenDoorsOrder or MoveTwoAcross or OpenTwoDoors or OpenRedBlue or PickupLoc or MiniBoss or wrapping or pickl" 2>&1 | tail -5 | tee $OUT/pytest_sentence.log
echo "== bosslevel bench" | tee $OUT/boss.txt
for f in 1 0; do timeout 200 python bench.py --workload bosslevel --fused $f --steps 256 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json bosslevel_fused$f; done 2>&1 | tee -a $OUT/boss.txt
timeout 200 python bench.py --workload bosslevel --envs-per-gpu 131072 --steps 256 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json bosslevel_131072 | tee -a $OUT/boss.txt
cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_boss -o boss -- python $ROOT/bench.py --workload bosslevel --steps 256 --warmup 64 --no-cpu-baseline > $OUT/prof_boss.log 2>&1
cp $(find $OUT/prof_boss -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bosslevel.csv; rm -rf $OUT/prof_boss; head -6 $OUT/kernel_stats_bosslevel.csv | cut -c1-160
