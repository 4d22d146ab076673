#!/bin/bash
# Round 5: BabyAI-GoTo x 131 072 with a deeper spare ring (the default 256 is halved to 128 by the 16 GB cap: 552 B per slot and env)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5n; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], d['config']['episodes_finished_rank0']))
except Exception as ex: print('$1 FAILED', ex)"; }
for cfg in "MG_SPARE_RING=64" "MG_SPARE_RING=128" "MG_SPARE_RING=256 MG_RING_CAP_GB=64" "MG_SPARE_RING=256 MG_RING_CAP_GB=64 MG_REFILL_WPS=2" "MG_SPARE_RING=256 MG_RING_CAP_GB=64 MG_REFILL_WPS=8"; do
  env $cfg python bench.py --workload babyai_goto --steps 320 --warmup 64 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 320 $cfg"
  env $cfg python bench.py --workload babyai_goto --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 1024 $cfg"
done | tee $OUT/goto_ring.txt
for cfg in "MG_SPARE_RING=128" "MG_SPARE_RING=256 MG_RING_CAP_GB=64"; do
  env $cfg python bench.py --workload bosslevel --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "bosslevel steps 1024 $cfg"
  env $cfg python bench.py --workload multiroom --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "multiroom steps 1024 $cfg"
done | tee -a $OUT/goto_ring.txt
