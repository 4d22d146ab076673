#!/bin/bash
# round 3: the cost of a launch on this box (profiles/tools/launch_floor.hip) beside k_roll7's one-step launch; SynthS5R2 redraw tests
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3s; mkdir -p $OUT
echo "== launch floor" | tee $OUT/launch_floor.txt
hipcc --offload-arch=gfx950 -O3 profiles/tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor 2>&1 | tee -a $OUT/launch_floor.txt
/tmp/launch_floor 2>&1 | tee -a $OUT/launch_floor.txt
echo "== SynthS5R2"
timeout 900 python -m pytest tests/test_gpu_synths5r2.py -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest_synths5r2.log
