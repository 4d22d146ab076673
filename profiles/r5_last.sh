#!/bin/bash
# Round 5, the LAST GPU call: exactly what the driver runs at round end, on the tree that ships -- the GPU suite (serial, -x), smoke(), the driver's bench line
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5last; mkdir -p $OUT
git -C $ROOT rev-parse HEAD 2>/dev/null | tee $OUT/head.txt
python -c "
from minigrid_amd import build; print('library stale:', build._stale(), ' step_kernel_srchash:', build.step_kernel_hash())" | tee $OUT/build_state.txt
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_gpu_serial.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_last.json 2> $OUT/bench_driver_last.err; cut -c1-700 $OUT/bench_driver_last.json
python - <<'PY' | tee $OUT/native_loaded.txt
import minigrid_amd as mg
e = mg.make_vec("MiniGrid-Empty-8x8-v0", 64); e.reset(seed=0); e.close()
print([l.split()[-1] for l in open("/proc/self/maps") if ".so" in l and ("minigrid" in l or "liboracle" in l)][:1])
PY
du -sh $ROOT
