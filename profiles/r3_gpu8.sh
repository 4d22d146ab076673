#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "pickl" > $OUT/pytest_pickle.log 2>&1; echo "rc=$?"; head -70 $OUT/pytest_pickle.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_roll.py -x -q > $OUT/pytest_roll.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_roll.log | cut -c1-200
