#!/bin/bash
# The inter-wave LDS protocol of k_roll7 under stress: tests/test_gpu_lds_protocol.py on the product library, then on a build whose
# step log has TWO entries (the dynamics wave waits on the encode waves in nearly every step) with the staged split's ring at two stagings as well.
#   build here first:  python profiles/variant_build.py log2 -DMG_ROLL_LOG_STEPS=2
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/${1:-r5protocol}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lds_protocol.py -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_protocol_product.log
LIB=$ROOT/minigrid_amd/libminigrid_hip_log2.so
[ -f $LIB ] || { echo "no stress build (python profiles/variant_build.py log2 -DMG_ROLL_LOG_STEPS=2)"; exit 1; }
MINIGRID_AMD_LIB=$LIB MG_DRING=2 timeout 900 python -m pytest tests/test_gpu_lds_protocol.py tests/test_gpu_roll.py tests/test_gpu_launch_lengths.py -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_protocol_log2_dring2.log
