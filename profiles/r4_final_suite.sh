#!/bin/bash
# the GPU suite + smoke() on the very last tree of the round (the full evidence run -- profiles/r4_final.sh -- predates two default changes
# that do not touch the four BASELINE workloads: the spare-ring depth of the wavefront-per-episode RoomGrid levels and an A/B switch)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4final2; mkdir -p $OUT
python -c "from minigrid_amd import build; print('library stale:', build._stale())" | tee $OUT/build_state.txt
timeout 900 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > $OUT/pytest_gpu_full_suite.log; tail -3 $OUT/pytest_gpu_full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
