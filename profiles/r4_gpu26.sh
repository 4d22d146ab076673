#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4z; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
A=$ROOT/minigrid_amd/libminigrid_hip_attr.so
for rep in 1 2; do
for x in 0 256 32; do MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty attr MG_EXP=$x (nt policy)"; done
for x in 0 256 32; do MG_NT_BYTES=-1 MG_EXP=$x MINIGRID_AMD_LIB=$A timeout 100 python bench.py --steps 2048 --warmup 256 --no-cpu-baseline 2>&1 | line "empty attr MG_EXP=$x (plain stores)"; done
done | tee $OUT/attribution_l2_window.txt
