#!/bin/bash
# round 4: the sentence levels split (stepping / verifying wave + one encode wave over one grid copy): parity, BossLevel
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4boss6; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
timeout 900 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider -k "Boss or sentence or GoToSeq or Synth or OpenTwoDoors or done_actions or MoveTwo or PickupLoc or synths5r2 or wrapping or pickling or dynobs" > $OUT/pytest_sentence.log 2>&1; echo "sentence + dynobs tests rc=$?" | tee $OUT/rc.txt
tail -6 $OUT/pytest_sentence.log
B="timeout 200 python bench.py --workload bosslevel --no-cpu-baseline --steps 512 --warmup 128"
for n in 131072 32768 262144; do
  $B --envs-per-gpu $n 2>&1 | line "bosslevel x $n split (2 waves)  "
  MG_SENT_SPLIT=0 $B --envs-per-gpu $n 2>&1 | line "bosslevel x $n one wave        "
done | tee $OUT/bosslevel_split.txt
MG_ROLL_NW=2 timeout 100 python bench.py --workload dynobs16x16 --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "dynobs16x16 NW=2 (split now) " | tee -a $OUT/bosslevel_split.txt
timeout 100 python bench.py --workload dynobs16x16 --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "dynobs16x16 NW=3 default " | tee -a $OUT/bosslevel_split.txt
