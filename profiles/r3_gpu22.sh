#!/bin/bash
# round 3: prologue with the independent loads issued together, pipelined quad encode in the one-step launch too
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3v; mkdir -p $OUT
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused.py tests/test_gpu_fused_full.py tests/test_gpu_synths5r2.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ragged or 4096_envs or (widened and (BossLevel-v0 or GoToObjMaze-v0 or KeyCorridorS6R3 or Dynamic-Obstacles-8x8 or PutNear-8x8 or MultiRoom-N6 or Memory or Fetch-8x8))" 2>&1 | tail -3 | tee $OUT/pytest_b.log
line() { python - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s n=%-7d %7.3f G steps/s %6.2f us/step frac %.3f" % (sys.argv[2], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== bench" | tee $OUT/bench_lines.txt
for rep in 1 2; do
  for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
    timeout 200 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$w"
  done
  for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do
    timeout 100 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "$w unfused"
  done
  timeout 100 python bench.py --workload empty8x8 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "empty8x8 driver-sized"
done 2>&1 | tee -a $OUT/bench_lines.txt
echo "== anatomy of the one-step launch" | tee $OUT/unfused_anatomy2.txt
for ex in 0 2 6 22 30; do
  MG_EXP=$ex timeout 100 python bench.py --workload empty8x8 --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json "empty8x8 unfused MG_EXP=$ex"
done 2>&1 | tee -a $OUT/unfused_anatomy2.txt
