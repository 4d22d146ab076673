#!/bin/bash
# round 4: FullyObs staged-codes split: parity, then LavaCrossing FullyObs
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4full; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
timeout 900 python -m pytest tests/test_gpu_launch_lengths.py tests/test_gpu_fused_full.py tests/test_gpu_roll.py tests/test_gpu_fused.py -q -n 4 -p no:cacheprovider -k "True or full or Full or Lava or same_step or FourRooms or launch_length" > $OUT/pytest_full.log 2>&1; echo "full-obs tests rc=$?" | tee $OUT/rc.txt
tail -8 $OUT/pytest_full.log
B="timeout 120 python bench.py --workload lavacrossing_full --no-cpu-baseline --steps 2048 --warmup 256"
$B 2>&1 | line "lavacrossing_full split NW=3 ring 2 (default) " | tee $OUT/lava_split.txt
MG_DRING=4 $B 2>&1 | line "lavacrossing_full split NW=3 ring 4 " | tee -a $OUT/lava_split.txt
MG_ROLL_NW=4 $B 2>&1 | line "lavacrossing_full split NW=4 ring 2 " | tee -a $OUT/lava_split.txt
MG_ROLL_NW=2 $B 2>&1 | line "lavacrossing_full split NW=2 ring 2 " | tee -a $OUT/lava_split.txt
MG_FULL_SPLIT=0 $B 2>&1 | line "lavacrossing_full time split (round 3 shape) " | tee -a $OUT/lava_split.txt
