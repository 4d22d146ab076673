#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4o; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) frac %.3f host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], r['frac'], (d['host_ms']-d['event_ms'])*1e3))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for w in gotoredball lavacrossing_full; do
  for x in 0 8 32 40 46 16; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w attr MG_EXP=$x "; done
  MG_ROLL_SPLIT=0 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w time split "
  MG_ROLL_SHADOWS=2 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w 2 shadows "
  MG_MAX_FUSED=16 timeout 100 python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w 16-step launches "
  for n in 16384 65536 131072; do timeout 100 python bench.py --workload $w --envs-per-gpu $n --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w $n envs "; done
done | tee $OUT/gotoredball_lava_attr.txt
