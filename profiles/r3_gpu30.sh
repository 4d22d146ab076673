#!/bin/bash
# round 3: the full GPU suite and smoke() on the last build of the round (after the final_obs / DynamicObstacles knob changes)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3ad; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu_full_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 100 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 600 $OUT/bench_driver.json
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 900 $OUT/bench_default.json
