#!/bin/bash
# round 2, GPU call 4: no global loads in the step loop: fused tests, bench sweep, counters
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_philox.py -x -q > $O/t_fused.log 2>&1; echo "fused+philox rc=$?" | tee -a $O/summary.txt
tail -4 $O/t_fused.log
for lpe in 1 4; do
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  for f in 1 0; do
    MG_LPE=$lpe timeout 300 python bench.py --workload $w --fused $f --steps 2000 --warmup 300 --no-cpu-baseline > $O/bench_${w}_f${f}_l$lpe.json 2> $O/bench_${w}_f${f}_l$lpe.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${w}_f${f}_l$lpe.json").read().strip().splitlines()[-1])
    print("${w} lpe=$lpe fused=$f", "%.3f G steps/s" % (d["value"]/1e9), "us/step %.2f" % d["roofline"]["avg_step_us"], "frac %.3f" % d["roofline"]["frac"])
except Exception as ex:
    print("${w} lpe=$lpe fused=$f FAILED", ex)
PY
  done
done
done | tee -a $O/summary.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver$i.json 2> $O/bench_driver.err; python -c "
import json; d=json.loads(open('$O/bench_driver$i.json').read().strip().splitlines()[-1]); print('driver-like', d['value']/1e9, d['ms_per_step'])" | tee -a $O/summary.txt; done
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fused.py --deselect tests/test_gpu_philox.py > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt
tail -4 $O/t_all.log
bash profiles/pmc_sq.sh r2d empty8x8 2>&1 | tee $O/sq_empty8x8.txt | tail -24
