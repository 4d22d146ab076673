#!/bin/bash
# round 3, GPU call 2: k_roll7 (time-split fused kernel of the 7x7 view) -- parity first, then the occupancy / split sweeps.
export TMPDIR=/tmp
OUT=gpurun_out/r3b; mkdir -p $OUT
echo "== first process of the box" | tee $OUT/first_process.log
timeout 180 python profiles/first_process.py >> $OUT/first_process.log 2>&1; echo "rc=$?" | tee -a $OUT/first_process.log
echo "== k_roll7 tests"
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused_full.py tests/test_gpu_fused.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_roll.log
bench() {   # name, env settings..., bench args
  local name=$1; shift
  env "${ENVS[@]}" timeout 120 python bench.py --no-cpu-baseline "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - $OUT/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %-22s n=%-7d %7.3f G steps/s %6.2f us/step  event %.3f ms host %.3f ms frac %.3f" % (sys.argv[2], d["config"]["env_id"][:22], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3, d["event_ms"], d["host_ms"], d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== sweeps" | tee $OUT/sweeps.txt
for nw in 1 2 4; do for n in 65536 262144; do ENVS=(MG_ROLL_NW=$nw); bench empty_nw${nw}_n$n --workload empty8x8 --envs-per-gpu $n --steps 1024 --warmup 128; done; done 2>&1 | tee -a $OUT/sweeps.txt
for r in 0.06 0.2 0.3; do ENVS=(MG_ROLL_NW=4 MG_ROLL_RATIO=$r); bench empty_nw4_ratio$r --workload empty8x8 --steps 1024 --warmup 128; done 2>&1 | tee -a $OUT/sweeps.txt
ENVS=(A=1)
for nw in 1 2 4; do ENVS=(MG_ROLL_NW=$nw); bench doorkey_nw$nw --workload doorkey8x8 --steps 1024 --warmup 128; bench gotoredball_nw$nw --workload gotoredball --steps 1024 --warmup 128; done 2>&1 | tee -a $OUT/sweeps.txt
ENVS=(A=1)
bench empty_unfused --workload empty8x8 --fused 0 --steps 512 --warmup 64 2>&1 | tee -a $OUT/sweeps.txt
bench doorkey_unfused --workload doorkey8x8 --fused 0 --steps 512 --warmup 64 2>&1 | tee -a $OUT/sweeps.txt
bench driver_sized --steps 20 --warmup 5 2>&1 | tee -a $OUT/sweeps.txt
bench lavacrossing_full --workload lavacrossing_full --steps 1024 --warmup 128 2>&1 | tee -a $OUT/sweeps.txt
echo "== kernel trace of the default fused run"
ROOT=$PWD
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_empty -o empty -- python $ROOT/bench.py --steps 512 --warmup 128 --no-cpu-baseline > $ROOT/$OUT/prof_empty.log 2>&1 )
cp $(find $OUT/prof_empty -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_empty8x8.csv 2>/dev/null; head -4 $OUT/kernel_stats_empty8x8.csv | cut -c1-200
rm -rf $OUT/prof_empty
echo "== full GPU suite"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_full.log
echo "== sanitizer retry"
RT=$(python profiles/asan_build.py --runtime)
if [ -f minigrid_amd/libminigrid_hip_asan.so ]; then
  HSA_XNACK=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 MINIGRID_AMD_LIB=$PWD/minigrid_amd/libminigrid_hip_asan.so timeout 120 python profiles/first_process.py 4096 > $OUT/asan_A.log 2>&1; echo "asan A (protect_shadow_gap=0) rc=$? $(tail -1 $OUT/asan_A.log | cut -c1-100)"
  HSA_XNACK=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 MINIGRID_AMD_NO_TORCH=1 MINIGRID_AMD_LIB=$PWD/minigrid_amd/libminigrid_hip_asan.so timeout 120 python profiles/first_process.py 4096 > $OUT/asan_B.log 2>&1; echo "asan B (rocm runtime, no torch) rc=$? $(tail -1 $OUT/asan_B.log | cut -c1-100)"
fi 2>&1 | tee $OUT/asan_retry.txt
