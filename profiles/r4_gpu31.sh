#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4ae; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f) host-event %.1f us' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], (d['host_ms']-d['event_ms'])*1e3))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for x in 0 8 32 46 110 174 238; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --workload gotoredball --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball attr MG_EXP=$x "; done | tee $OUT/gotoredball_attr_final.txt
for n in 65536 131072; do timeout 100 python bench.py --workload gotoredball --envs-per-gpu $n --steps 2048 --warmup 512 --no-cpu-baseline 2>&1 | line "gotoredball $n envs "; done | tee -a $OUT/gotoredball_attr_final.txt
for x in 0 32 46 110; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --workload lavacrossing_full --steps 4096 --warmup 512 --no-cpu-baseline 2>&1 | line "lavacrossing_full attr MG_EXP=$x "; done | tee -a $OUT/gotoredball_attr_final.txt
