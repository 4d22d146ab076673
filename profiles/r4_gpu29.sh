#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4ac; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_roll.py -x -q -m gpu -n 4 2>&1 | tail -12 | tee $OUT/pytest_same_step_sentence.log
