#!/bin/bash
# RGB observation path (k_step tile map + k_render): bench lines, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE PMC
# passes (separate runs) and the box's plain fill bandwidth for reference.
# usage (via gpurun): bash profiles/collect_rgb.sh <tag>
TAG=${1:-r1}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for w in empty8x8_rgb doorkey8x8_rgb_partial; do
  extra="--no-cpu-baseline"; [ $w = empty8x8_rgb ] && extra=""
  timeout 120 python bench.py --workload $w --steps 300 --warmup 30 $extra > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - $OUT/bench_$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["env_id"], d["config"]["obs_mode"], "%.3f G steps/s"%(d["value"]/1e9), "%.2f us/step"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"])
PY
done
python - > $OUT/render_fill_ceiling.txt <<'PY'
import torch
x = torch.empty(805306368, dtype=torch.uint8, device="cuda")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
us = t(lambda: x.fill_(7)); print(f"torch fill_ of 805306368 B (one empty8x8_rgb step's frames): {us:.1f} us = {805.306368e6/us/1e6:.2f} TB/s written")
PY
cat $OUT/render_fill_ceiling.txt
cd /tmp
for w in empty8x8_rgb doorkey8x8_rgb_partial; do
  CMD="python $ROOT/bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline"
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- $CMD > $OUT/prof_$w.log 2>&1
  cp $(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$w.csv
  rm -rf $OUT/prof_$w
  head -3 $OUT/kernel_stats_$w.csv | cut -c1-160
done
w=empty8x8_rgb
CMD="python $ROOT/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$w -o $w -- $CMD > $OUT/pmc_${c}_$w.log 2>&1
  python $ROOT/profiles/summarize_pmc.py $(find $OUT/pmc_${c}_$w -name '*counter_collection.csv' | head -1) > $OUT/pmc_${c}_$w.txt
  cat $OUT/pmc_${c}_$w.txt
  rm -rf $OUT/pmc_${c}_$w
done
