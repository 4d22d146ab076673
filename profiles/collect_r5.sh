#!/bin/bash
# Round-5 evidence on the GPU box: for each (workload, launch shape) a `rocprofv3 --kernel-trace --stats` pass and -- separately, as
# MI355X_MICROARCH.md prescribes -- a `--pmc FETCH_SIZE` and a `--pmc WRITE_SIZE` pass of the SAME bench.py command, each with the run's
# parameters in meta_<workload><suffix>.json so that bench.py only quotes counters taken at ITS batch size and steps per launch:
#   suffix ""       bench.py --workload W --steps 512 --warmup 128      (32-step launches, steady state)
#   suffix "_spl20" bench.py --workload W --steps 400 --warmup 20 --spl 20   (the driver's launch shape -- 20 steps per launch -- over TWENTY launches)
# every meta carries step_kernel_srchash (minigrid_amd/build.py): bench.py quotes a pass only on the build of the step kernels it measured
# usage (via gpurun): bash profiles/collect_r4.sh <tag> "<shapes: long spl20>" <workloads...>
TAG=${1:-r5}; SHAPES=${2:-"long spl20"}; shift; shift
PROF_WL=${@:-empty8x8}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $ROOT/profiles/r5
cd /tmp
for w in $PROF_WL; do for shape in $SHAPES; do
  if [ $shape = spl20 ]; then SFX=_spl20; ARGS="--steps 400 --warmup 20 --spl 20"; else SFX=""; ARGS="--steps 512 --warmup 128"; fi
  CMD="python $ROOT/bench.py --gpus 1 --workload $w $ARGS --no-cpu-baseline"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w$SFX -o $w -- $CMD > $OUT/prof_$w$SFX.log 2>&1
  cp $(find $OUT/prof_$w$SFX -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$w$SFX.csv
  python - $(find $OUT/prof_$w$SFX -name '*kernel_trace.csv' | head -1) $OUT/prof_$w$SFX.log $OUT/meta_$w$SFX.json $w "$ARGS" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_roll7" in r["Kernel_Name"] or "k_step" in r["Kernel_Name"]]
dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
spl = line["config"]["steps_per_launch"]
# the FULL launches (spl steps): everything longer than half the MEDIAN duration (the full launches are the majority; a one-step reset observation is a
# tenth of one).  (Round 4's "0.7 x the longest" dropped every normal launch of a run that held one slow outlier.)
full = [d for d in dur if d > 0.5 * dur[len(dur) // 2]]
meta = {"workload": sys.argv[4], "envs_per_gpu": line["config"]["envs_per_gpu"], "steps_per_launch": spl,
        "full_launches": len(full), "full_launch_filter": "> 0.5 x median duration", "full_launch_avg_us": sum(full) / len(full) / 1e3, "full_launch_max_us": full[-1] / 1e3,
        "all_step_kernel_launches_us": [d / 1e3 for d in dur],
        "library_build": line["config"].get("library_build"), "step_kernel_srchash": line["config"]["step_kernel_srchash"], "environment": line["config"].get("environment"),
        "command": "bench.py --gpus 1 --workload %s %s --no-cpu-baseline under rocprofv3 --kernel-trace --stats" % (sys.argv[4], sys.argv[5])}
if len(meta["all_step_kernel_launches_us"]) > 40: del meta["all_step_kernel_launches_us"]
json.dump(meta, open(sys.argv[3], "w"), indent=1)
print(meta)
PY
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$w$SFX -o $w -- $CMD > $OUT/pmc_${c}_$w$SFX.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find $OUT/pmc_${c}_$w$SFX -name '*counter_collection.csv' | head -1) > $OUT/pmc_${c}_$w$SFX.txt
    grep -E "k_roll7|k_step" $OUT/pmc_${c}_$w$SFX.txt
  done
  rm -rf $OUT/prof_$w$SFX $OUT/pmc_FETCH_SIZE_$w$SFX $OUT/pmc_WRITE_SIZE_$w$SFX
  # the bench lines below quote kernel time / HBM traffic from profiles/r5: give them THIS build's (the box's copy of the tree is scratch)
  cp $OUT/kernel_stats_$w$SFX.csv $OUT/meta_$w$SFX.json $OUT/pmc_FETCH_SIZE_$w$SFX.txt $OUT/pmc_WRITE_SIZE_$w$SFX.txt $ROOT/profiles/r5/
  head -3 $OUT/kernel_stats_$w$SFX.csv | cut -c1-160
done; done
cd $ROOT
