#!/bin/bash
# round 4: the new kernels under the red-zone guard (every device buffer between two pattern-filled 4 KB zones, checked at mg_sync / mg_destroy)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4guard; mkdir -p $OUT
MG_GUARD=1 timeout 900 python -m pytest tests/test_gpu_dynobs.py tests/test_gpu_launch_lengths.py tests/test_gpu_roll.py -q -n 4 -p no:cacheprovider > $OUT/pytest_guarded.log 2>&1; echo "guarded tests rc=$?" | tee $OUT/rc.txt
tail -4 $OUT/pytest_guarded.log
