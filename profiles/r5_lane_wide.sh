#!/bin/bash
# Round 5, first GPU call for the MG_LANE_WIDE variant (mg_genlane.h; written in round 4 after the GPU budget was spent, never run on a GPU):
# the lane-per-episode refill for every level (DynamicObstacles draws inside its step kernel).  Its generators are the ones tests/test_generators_cpu.py pins on the CPU; what a GPU
# has to show is (1) the kernels run and the whole GPU parity suite stays green with the variant library, (2) what it buys per level
# (MG_LANE_GEN=0 in the SAME library = the wavefront-per-episode refill).  If both hold: MG_LANE_WIDE becomes the default (one line, mg_genlane.h).
#   build here first:  python profiles/variant_build.py lanewide --units=mg_gen_lane.hip,mg_api.hip -DMG_LANE_WIDE=1
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5lanewide; mkdir -p $OUT
LIB=$ROOT/minigrid_amd/libminigrid_hip_lanewide.so
[ -f $LIB ] || { echo "build the variant first (see the header)"; exit 1; }
export MINIGRID_AMD_LIB=$LIB
python -c "from minigrid_amd import _binding as B; print(B.load().mg_build_info().decode())" | tee $OUT/build_info.txt
# (1) parity: every GPU test that resets or refills a level the variant moves to the lane kernels (the whole suite is the bar; these first)
MG_GUARD=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roll.py tests/test_gpu_philox.py tests/test_gpu_synths5r2.py -q -m gpu -n 4 -x 2>&1 | tail -8 | tee $OUT/pytest_parity_guarded.log
timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 | tee $OUT/pytest_gpu_full_suite.log
# (2) what it buys: generator-bound levels, lane refill against the wavefront-per-episode refill of the same library
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.2f us/step' % (d['value']/1e9, d['ms_per_step']*1e3))"; }
for w in keycorridor unlockpickup babyai_goto bosslevel multiroom; do
  for lg in 1 0; do
    MG_LANE_GEN=$lg timeout 300 python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_${w}_lane$lg.json | line "$w MG_LANE_GEN=$lg"
  done
done | tee $OUT/bench_lines.txt
# the refill kernels' own durations
cd /tmp && MG_LANE_GEN=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_keycorridor -o kc -- python $ROOT/bench.py --workload keycorridor --steps 256 --warmup 32 --no-cpu-baseline > /dev/null 2>&1
grep -h "k_refill\|k_roll7" $OUT/prof_keycorridor/*kernel_stats.csv 2>/dev/null | cut -c1-200 | tee $OUT/kernel_stats_keycorridor.txt
