#!/bin/bash
# Round 6: the inter-wave spins of k_roll7 BOUNDED (VERDICT r5 "next" #7) -- a protocol regression as a failing test, not a hung lease.
#   build here first:
#     python profiles/variant_build.py spinbound -DMG_SPIN_BOUND=16777216
#     python profiles/variant_build.py spinneg -DMG_SPIN_BOUND=65536 -DMG_SPIN_NEGATIVE_CONTROL
# 1. the protocol stress test on the bounded build: green, i.e. no wait of a full-size run comes anywhere near 2^24 polls
# 2. the negative control (an encode wave that never reports progress): the launch ENDS and the next sync raises -- the bound fires instead of a hang
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/${1:-r6}; mkdir -p $OUT
LIB=$ROOT/minigrid_amd/libminigrid_hip_spinbound.so
[ -f $LIB ] || { echo "no bounded build"; exit 1; }
MINIGRID_AMD_LIB=$LIB timeout 900 python -m pytest tests/test_gpu_lds_protocol.py -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_lds_protocol_spin_bound.log
NEG=$ROOT/minigrid_amd/libminigrid_hip_spinneg.so
[ -f $NEG ] && MINIGRID_AMD_LIB=$NEG timeout 300 python - <<'PY' 2>&1 | tail -5 | tee $OUT/spin_bound_negative_control.log
import time
import minigrid_amd as mg
env = mg.make_vec("MiniGrid-Empty-8x8-v0", 4096)
env.reset(seed=0)
t0 = time.time()
try:
    env.rollout(32, action_seed=1, fused=True)       # the log split: the dynamics wave waits for progress nobody reports
    env.sync()
    print("NEGATIVE CONTROL FAILED: the broken protocol went unnoticed")
except Exception as ex:
    print("negative control ok: the launch ended after %.2f s and the sync raised: %s" % (time.time() - t0, str(ex)[:160]))
PY
