#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4ab; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k done_actions 2>&1 | tail -12 | tee $OUT/pytest_done_actions.log
timeout 900 python -m pytest tests/test_gpu_roll.py tests/test_gpu_launch_lengths.py -x -q -m gpu -n 4 2>&1 | tail -3 | tee -a $OUT/pytest_done_actions.log
