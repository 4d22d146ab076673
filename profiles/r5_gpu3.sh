#!/bin/bash
# Round 5, GPU call 3: (1) why the MG_LANE_WIDE variant failed 192 GPU tests (tracebacks), (2) the inter-wave LDS protocol under stress,
# (3) cache-policy A/B of the observation stream (buffer stores with sc1 / sc0 sc1 / sc1 nt / nt / plain bits) + a 16-entry step log,
# (4) where the time goes on the generator-bound families (kernel trace).
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5c; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) frac %.3f' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac']))
except Exception as ex: print('$1 FAILED', ex)"; }
# ---- (1)
LW=$ROOT/minigrid_amd/libminigrid_hip_lanewide.so
if [ -f $LW ]; then
  MINIGRID_AMD_LIB=$LW timeout 300 python -m pytest "tests/test_gpu_parity.py::test_generators_match_reference_goldens[MiniGrid-Empty-8x8-v0]" "tests/test_gpu_parity.py::test_rgb_frames_match_reference_goldens[partial-rgb_MiniGrid-FourRooms-v0]" -q -m gpu --tb=short 2>&1 | tail -60 > $OUT/lanewide_tracebacks.log
  MINIGRID_AMD_LIB=$LW timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --tb=line 2>&1 | tail -15 > $OUT/lanewide_parity_serial.log
  tail -5 $OUT/lanewide_parity_serial.log
fi
# ---- (2)
bash profiles/r5_protocol_stress.sh r5c
# ---- (3)
for rep in 1 2; do
for v in "" aux0 aux2 aux16 aux17 aux18 log16; do
  LIB=""; [ -n "$v" ] && LIB=$ROOT/minigrid_amd/libminigrid_hip_$v.so
  [ -n "$v" ] && [ ! -f $LIB ] && continue
  MINIGRID_AMD_LIB=$LIB timeout 200 python bench.py --workload empty8x8 --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "empty8x8 ${v:-product}"
  MINIGRID_AMD_LIB=$LIB timeout 200 python bench.py --workload doorkey8x8 --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "doorkey8x8 ${v:-product}"
done; done | tee $OUT/ab_store_policy.txt
for v in "" aux16 aux17; do
  LIB=""; [ -n "$v" ] && LIB=$ROOT/minigrid_amd/libminigrid_hip_$v.so
  [ -n "$v" ] && [ ! -f $LIB ] && continue
  MINIGRID_AMD_LIB=$LIB timeout 200 python bench.py --workload lavacrossing_full --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "lavacrossing_full ${v:-product}"
  MINIGRID_AMD_LIB=$LIB timeout 200 python bench.py --workload gotoredball --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "gotoredball ${v:-product}"
  MINIGRID_AMD_LIB=$LIB timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "empty8x8 driver-shape ${v:-product}"
done | tee -a $OUT/ab_store_policy.txt
# ---- (4)
cd /tmp
for w in bosslevel babyai_goto multiroom; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $ROOT/bench.py --workload $w --steps 256 --warmup 64 --no-cpu-baseline > $OUT/prof_$w.log 2>&1
  f=$(find $OUT/prof_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { cut -c1-170 $f | head -8 > $OUT/kernel_stats_$w.txt; cat $OUT/kernel_stats_$w.txt; }
  rm -rf $OUT/prof_$w
done
cd $ROOT
