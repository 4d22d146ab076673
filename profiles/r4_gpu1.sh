#!/bin/bash
# round 4, call 1: (a) the new launch-length parity tests + the fused / roll subsets on the product build (MG_EXP compiled out),
# (b) the write-stream microbenchmark, (c) A/B of the store-side variants, (d) the driver-sized PMC / kernel-trace passes, (e) driver-sized lines
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4a; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())"
timeout 900 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_launch_lengths.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rollstore.hip -o /tmp/rollstore && timeout 120 /tmp/rollstore | tee $OUT/rollstore.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f  8d-frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], r['survey_8d']['frac']))"; }
for rep in 1 2; do
  timeout 100 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "product        "
  for v in thr0 thr2 thr4 xcd; do
    MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_$v.so timeout 100 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "variant $v   "
  done
  for x in 0 32 2 6 22; do
    MG_EXP=$x MINIGRID_AMD_LIB=$ROOT/minigrid_amd/libminigrid_hip_attr.so timeout 100 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "attr MG_EXP=$x  "
  done
done | tee $OUT/ab_store_variants.txt
for v in product xcd thr2; do
  L=$ROOT/minigrid_amd/libminigrid_hip_$v.so; [ $v = product ] && L=$ROOT/minigrid_amd/libminigrid_hip.so
  MINIGRID_AMD_LIB=$L timeout 100 python bench.py --workload doorkey8x8 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "doorkey $v   "
done | tee -a $OUT/ab_store_variants.txt
bash profiles/collect_r4.sh r4a "spl20 long" empty8x8 2>&1 | tail -12
for i in 1 2 3; do timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver$i.json 2> $OUT/bench_driver$i.err; python - $OUT/bench_driver$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("driver-sized %.3f G  host %.1f us  event %.1f us  incl-dev-sync %.1f us  frac %.3f traffic %s kernel_us_per_step %s" % (d["value"]/1e9, d["host_ms"]*1e3, d["event_ms"]*1e3, d["host_ms_incl_device_sync"]*1e3, r["frac"], r["traffic"], r["kernel_us_per_step"]))
PY
done | tee $OUT/driver_lines.txt
