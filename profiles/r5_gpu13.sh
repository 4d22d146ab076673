#!/bin/bash
# Round 5: what bounds BabyAI-GoTo x 131 072 -- windows with and without the synchronized truncation burst (max_steps 576), refill geometry
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5m; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes %d' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], d['config']['episodes_finished_rank0']))
except Exception as ex: print('$1 FAILED', ex)"; }
for w in babyai_goto multiroom keycorridor; do
  python bench.py --workload $w --steps 320 --warmup 64 --no-cpu-baseline 2>/dev/null | line "$w steps 320 warmup 64 (no truncation burst inside for max_steps 576)"
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 warmup 128"
  python bench.py --workload $w --steps 2304 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 2304 warmup 128"
done | tee $OUT/goto_windows.txt
for wps in 1 2 4 8; do MG_REFILL_WPS=$wps python bench.py --workload babyai_goto --steps 320 --warmup 64 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 320 MG_REFILL_WPS=$wps"; done | tee -a $OUT/goto_windows.txt
MG_LANE_GEN=0 python bench.py --workload babyai_goto --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto 1024 MG_LANE_GEN=0 (direct generation by wavefronts too)" | tee -a $OUT/goto_windows.txt
