#!/bin/bash
# Round-6 evidence on the GPU box: for each (workload, launch shape) a `rocprofv3 --kernel-trace --stats` pass and -- separately, as
# MI355X_MICROARCH.md prescribes -- a `--pmc FETCH_SIZE`, a `--pmc WRITE_SIZE` and an SQ pass (SQ_INSTS_VALU ...: bench.py's valu_issue_frac) of the
# SAME bench.py command, each with the run's parameters in meta_<workload><suffix>.json so that bench.py only quotes counters taken at ITS batch size
# and steps per launch on ITS build of the step kernels (step_kernel_srchash):
#   suffix ""       bench.py --workload W --steps 512 --warmup 128           (32-step launches)
#   suffix "_spl20" bench.py --workload W --steps 400 --warmup 20 --spl 20   (the driver's launch shape -- 20 steps per launch -- over TWENTY launches)
# bench.py's untimed de-phasing pre-roll (32 short launches + 32 reset observations) is in every trace: the launches of the TIMED region are the LAST
# ceil(steps / steps_per_launch) step-kernel dispatches of the run, and those are what full_launch_avg_us averages.
# usage (via gpurun): bash profiles/collect_r6.sh <tag> "<shapes: long spl20>" <workloads...>
TAG=${1:-r6}; SHAPES=${2:-"long spl20"}; shift; shift
PROF_WL=${@:-empty8x8}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $ROOT/profiles/r6
cd /tmp
for w in $PROF_WL; do for shape in $SHAPES; do
  if [ $shape = spl20 ]; then SFX=_spl20; ARGS="--steps 400 --warmup 20 --spl 20"; else SFX=""; ARGS="--steps 512 --warmup 128"; fi
  CMD="python $ROOT/bench.py --gpus 1 --workload $w $ARGS --no-cpu-baseline"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w$SFX -o $w -- $CMD > $OUT/prof_$w$SFX.log 2>&1
  cp $(find $OUT/prof_$w$SFX -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$w$SFX.csv
  python - $(find $OUT/prof_$w$SFX -name '*kernel_trace.csv' | head -1) $OUT/prof_$w$SFX.log $OUT/meta_$w$SFX.json $w "$ARGS" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_roll7" in r["Kernel_Name"] or "k_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
spl, steps = line["config"]["steps_per_launch"], line["steps"]
n_launch = -(-steps // spl)
timed = rows[-n_launch:]                                  # the timed region's launches: the last ones of the run (the pre-roll and the warm-up come before)
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
meta = {"workload": sys.argv[4], "envs_per_gpu": line["config"]["envs_per_gpu"], "steps_per_launch": spl,
        "full_launches": len(dur), "full_launch_filter": "the last ceil(steps / steps_per_launch) step-kernel dispatches of the run = the timed region",
        "full_launch_avg_us": sum(dur) / len(dur) / 1e3, "full_launch_max_us": max(dur) / 1e3, "full_launch_min_us": min(dur) / 1e3,
        "timed_region_kernels": sorted({r["Kernel_Name"].split("(")[0] for r in timed}),
        "timed_launches_us": [d / 1e3 for d in dur],
        "episodes_finished_in_timed_region": line["config"].get("episodes_finished_in_timed_region_rank0"), "dephase": line["config"].get("dephase"),
        "library_build": line["config"].get("library_build"), "step_kernel_srchash": line["config"]["step_kernel_srchash"], "environment": line["config"].get("environment"),
        "command": "bench.py --gpus 1 --workload %s %s --no-cpu-baseline under rocprofv3 --kernel-trace --stats" % (sys.argv[4], sys.argv[5])}
if len(meta["timed_launches_us"]) > 40: del meta["timed_launches_us"]
json.dump(meta, open(sys.argv[3], "w"), indent=1)
print(meta)
PY
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$w$SFX -o $w -- $CMD > $OUT/pmc_${c}_$w$SFX.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find $OUT/pmc_${c}_$w$SFX -name '*counter_collection.csv' | head -1) > $OUT/pmc_${c}_$w$SFX.txt
    grep -E "k_roll7|k_step" $OUT/pmc_${c}_$w$SFX.txt
  done
  i=0; : > $OUT/sq_counters_$w$SFX.txt
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_IFETCH SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq${i}_$w$SFX -o x -- $CMD > $OUT/sq${i}_$w$SFX.log 2>&1
    python $ROOT/profiles/summarize_pmc.py $(find $OUT/sq${i}_$w$SFX -name '*counter_collection.csv' | head -1) | grep -E "k_roll7|k_step" | cut -c1-200 >> $OUT/sq_counters_$w$SFX.txt
    rm -rf $OUT/sq${i}_$w$SFX
  done
  grep SQ_INSTS_VALU $OUT/sq_counters_$w$SFX.txt | head -2
  rm -rf $OUT/prof_$w$SFX $OUT/pmc_FETCH_SIZE_$w$SFX $OUT/pmc_WRITE_SIZE_$w$SFX
  # the bench lines below quote kernel time / HBM traffic / VALU issue from profiles/r6: give them THIS build's (the box's copy of the tree is scratch)
  cp $OUT/kernel_stats_$w$SFX.csv $OUT/meta_$w$SFX.json $OUT/pmc_FETCH_SIZE_$w$SFX.txt $OUT/pmc_WRITE_SIZE_$w$SFX.txt $OUT/sq_counters_$w$SFX.txt $ROOT/profiles/r6/
  head -3 $OUT/kernel_stats_$w$SFX.csv | cut -c1-160
done; done
cd $ROOT
