#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3e; mkdir -p $OUT
echo "== debug: the failing time-split test"
timeout 120 python profiles/dbg_roll.py MiniGrid-DoorKey-8x8-v0 1001 3 1 2>&1 | tail -40 | tee $OUT/dbg_roll_nw1.txt
timeout 120 python profiles/dbg_roll.py MiniGrid-DoorKey-8x8-v0 1001 3 4 2>&1 | tail -12 | tee $OUT/dbg_roll_nw4.txt
timeout 120 python profiles/dbg_roll.py MiniGrid-DoorKey-8x8-v0 1024 640 1 2>&1 | tail -12 | tee $OUT/dbg_roll_long.txt
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_fused.py tests/test_gpu_fused_full.py tests/test_gpu_multi.py -q 2>&1 | tail -25 | tee $OUT/pytest_subset.log
echo "== sanitizer, torch-free process with the ROCm runtime"
RT=$(python profiles/asan_build.py --runtime)
HSA_XNACK=1 LD_PRELOAD=$RT LD_LIBRARY_PATH=/opt/rocm/lib ASAN_OPTIONS=detect_leaks=0 MINIGRID_AMD_NO_TORCH=1 MINIGRID_AMD_LIB=$PWD/minigrid_amd/libminigrid_hip_asan.so timeout 300 python profiles/first_process.py 4096 > $OUT/asan_C.log 2>&1; echo "asan C rc=$? $(tail -1 $OUT/asan_C.log | cut -c1-100)"; head -30 $OUT/asan_C.log | cut -c1-200
