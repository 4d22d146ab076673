#!/bin/bash
# round 2, GPU call 2: fused tests, bench lines, kernel-trace stats and SQ counters of the fused kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_fused.py -x -q > $O/t_fused.log 2>&1; echo "fused rc=$?" | tee -a $O/summary.txt
tail -4 $O/t_fused.log
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  for f in 1 0; do
    timeout 300 python bench.py --workload $w --fused $f --steps 2000 --warmup 300 --no-cpu-baseline > $O/bench_${w}_f$f.json 2> $O/bench_${w}_f$f.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${w}_f$f.json").read().strip().splitlines()[-1])
    print("${w} fused=$f", "%.3f G steps/s" % (d["value"]/1e9), "us/step %.2f" % d["roofline"]["avg_step_us"], "frac %.3f" % d["roofline"]["frac"])
except Exception as ex:
    print("${w} fused=$f FAILED", ex)
PY
  done
done | tee -a $O/summary.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; python -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]); print('driver-like', d['value']/1e9, d['ms_per_step'])" | tee -a $O/summary.txt
for w in empty8x8 gotoredball doorkey8x8; do
  bash profiles/kstats.sh $w 2>&1 | tee $O/kstats_$w.txt | tail -12
done
bash profiles/pmc_sq.sh r2b empty8x8 2>&1 | tee $O/sq_empty8x8.txt | tail -8
bash profiles/pmc_sq.sh r2b doorkey8x8 2>&1 | tee $O/sq_doorkey8x8.txt | tail -8
