#!/bin/bash
# Round 6, the last call: exactly what the driver runs at round end on a fresh box -- the GPU suite serially, smoke(), the bench line
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6last; mkdir -p $OUT
( time python -m pytest tests/ -x -q -m gpu ) 2>&1 | tail -6 | tee $OUT/pytest_gpu_serial.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-700 $OUT/bench_driver.json
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
