#!/bin/bash
# Collect the per-round evidence on the GPU box: bench lines for the 4 BASELINE workloads, rocprofv3 kernel stats
# and (separate passes, as MI355X_MICROARCH.md prescribes) the FETCH_SIZE / WRITE_SIZE PMC counters.
# usage (via gpurun): bash profiles/collect.sh <tag> [workloads for rocprof...]
TAG=${1:-r1}; shift
PROF_WL=${@:-empty8x8 gotoredball}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  extra="--no-cpu-baseline"; [ $w = empty8x8 ] && extra=""
  timeout 300 python bench.py --workload $w --steps 2000 --warmup 300 $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - $OUT/bench_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["env_id"], "%.3f G steps/s"%(d["value"]/1e9), "%.2f us/step"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"])
PY
done
cd /tmp
for w in $PROF_WL; do
  CMD="python $ROOT/bench.py --workload $w --steps 512 --warmup 128 --no-cpu-baseline"   # whole fused launches only
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- $CMD > $OUT/prof_$w.log 2>&1
  cp $(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$w.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$w -o $w -- $CMD > $OUT/pmc_${c}_$w.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find $OUT/pmc_${c}_$w -name '*counter_collection.csv' | head -1) > $OUT/pmc_${c}_$w.txt
    cat $OUT/pmc_${c}_$w.txt
  done
  rm -rf $OUT/prof_$w $OUT/pmc_*_$w
  head -3 $OUT/kernel_stats_$w.csv | cut -c1-160
done
