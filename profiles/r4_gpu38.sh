#!/bin/bash
# round 4: spare staging on / off (MG_ROLL_SHADOWS): the big-grid sentence levels are LDS-bound at two single-wave workgroups per CU
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4boss; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
for w in bosslevel; do
  for sh in 1 0; do
    MG_ROLL_SHADOWS=$sh timeout 200 python bench.py --workload $w --no-cpu-baseline --steps 512 --warmup 128 2>&1 | line "$w x 131072 shadows=$sh "
    MG_ROLL_SHADOWS=$sh timeout 200 python bench.py --workload $w --envs-per-gpu 32768 --no-cpu-baseline --steps 512 --warmup 128 2>&1 | line "$w x 32768 shadows=$sh "
  done
done | tee $OUT/shadows_bosslevel.txt
for w in doorkey8x8 gotoredball lavacrossing_full; do
  for sh in 1 0; do MG_ROLL_SHADOWS=$sh timeout 200 python bench.py --workload $w --no-cpu-baseline --steps 2048 --warmup 256 2>&1 | line "$w shadows=$sh "; done
done | tee -a $OUT/shadows_bosslevel.txt
