#!/bin/bash
# round 3: SAME_STEP with final_obs (by composition: NEXT_STEP step + masked reset of the finished envs)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3aa; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_roll.py -m gpu -q -k "same_step" 2>&1 | tail -30 | tee $OUT/pytest_final_obs.log
