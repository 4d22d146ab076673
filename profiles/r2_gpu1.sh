#!/bin/bash
# round 2, GPU call 1: new fused/ring tests first (fail fast), then the whole GPU suite, then bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/r2a/t_fused.log 2>&1; echo "fused rc=$?" | tee -a gpurun_out/r2a/summary.txt
tail -5 gpurun_out/r2a/t_fused.log
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fused.py > gpurun_out/r2a/t_all.log 2>&1; echo "all rc=$?" | tee -a gpurun_out/r2a/summary.txt
tail -5 gpurun_out/r2a/t_all.log
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  for f in 1 0; do
    timeout 300 python bench.py --workload $w --fused $f --steps 2000 --warmup 300 --no-cpu-baseline > gpurun_out/r2a/bench_${w}_f$f.json 2> gpurun_out/r2a/bench_${w}_f$f.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2a/bench_${w}_f$f.json").read().strip().splitlines()[-1])
    print("${w} fused=$f", "%.3f G steps/s" % (d["value"]/1e9), "us/step %.2f" % d["roofline"]["avg_step_us"], "frac %.3f" % d["roofline"]["frac"])
except Exception as ex:
    print("${w} fused=$f FAILED", ex)
PY
  done
done | tee -a gpurun_out/r2a/summary.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_driver.json 2> gpurun_out/r2a/bench_driver.err; tail -c 1500 gpurun_out/r2a/bench_driver.json
