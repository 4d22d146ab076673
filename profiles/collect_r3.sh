#!/bin/bash
# Round-3 evidence on the GPU box: bench lines of the BASELINE workloads, rocprofv3 kernel stats and (separate passes, as
# MI355X_MICROARCH.md prescribes) the FETCH_SIZE / WRITE_SIZE PMC counters, each with the run's parameters in meta_<workload>.json so
# that bench.py only quotes counters taken at ITS batch size and steps per launch.
# Order: the profiler passes first, then the bench lines (which quote them).
# usage (via gpurun): bash profiles/collect_r3.sh <tag> [workloads for rocprof...]
TAG=${1:-r3}; shift
PROF_WL=${@:-empty8x8 doorkey8x8 lavacrossing_full gotoredball}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
for w in $PROF_WL; do
  CMD="python $ROOT/bench.py --workload $w --steps 512 --warmup 128 --no-cpu-baseline"   # whole fused launches only
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- $CMD > $OUT/prof_$w.log 2>&1
  cp $(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$w.csv
  python - $(find $OUT/prof_$w -name '*kernel_trace.csv' | head -1) $OUT/prof_$w.log $OUT/meta_$w.json $w <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_roll7" in r["Kernel_Name"] or "k_step" in r["Kernel_Name"]]
dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
full = [d for d in dur if d > 0.5 * dur[-1]]                     # the full fused launches (the rest: one-step reset observations)
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
meta = {"workload": sys.argv[4], "envs_per_gpu": line["config"]["envs_per_gpu"], "steps_per_launch": line["config"]["steps_per_launch"],
        "full_launches": len(full), "full_launch_avg_us": sum(full) / len(full) / 1e3, "full_launch_max_us": full[-1] / 1e3,
        "command": "bench.py --workload %s --steps 512 --warmup 128 --no-cpu-baseline under rocprofv3 --kernel-trace --stats" % sys.argv[4]}
json.dump(meta, open(sys.argv[3], "w"), indent=1)
print(meta)
PY
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$w -o $w -- $CMD > $OUT/pmc_${c}_$w.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find $OUT/pmc_${c}_$w -name '*counter_collection.csv' | head -1) > $OUT/pmc_${c}_$w.txt
    grep -E "k_roll7|k_step" $OUT/pmc_${c}_$w.txt
  done
  rm -rf $OUT/prof_$w $OUT/pmc_*_$w
  # the bench lines below quote kernel time / HBM traffic from profiles/r3: give them THIS build's (the box's copy of the tree is scratch)
  cp $OUT/kernel_stats_$w.csv $OUT/meta_$w.json $OUT/pmc_FETCH_SIZE_$w.txt $OUT/pmc_WRITE_SIZE_$w.txt $ROOT/profiles/r3/
  head -3 $OUT/kernel_stats_$w.csv | cut -c1-160
done
cd $ROOT
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball; do
  extra="--no-cpu-baseline"; [ $w = empty8x8 ] && extra=""
  timeout 300 python bench.py --workload $w --steps 2048 --warmup 256 $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - $OUT/bench_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["env_id"], "%.3f G steps/s"%(d["value"]/1e9), "%.2f us/step"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"])
PY
done
for i in 1 2 3; do timeout 100 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver$i.json 2> $OUT/bench_driver$i.err; done
