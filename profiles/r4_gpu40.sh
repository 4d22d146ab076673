#!/bin/bash
# round 4: the sentence levels' instruction records staged in LDS for the launch (k_roll7<GG_SENTENCE>): parity, then BossLevel
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4boss3; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
timeout 900 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider -k "Boss or sentence or GoToSeq or Synth or OpenTwoDoors or done_actions or MoveTwo or PickupLoc or synths5r2 or wrapping or pickling" > $OUT/pytest_sentence.log 2>&1; echo "sentence tests rc=$?" | tee $OUT/rc.txt
tail -6 $OUT/pytest_sentence.log
for epw in 64 32; do for n in 131072 32768; do
  MG_ROLL_EPW=$epw timeout 200 python bench.py --workload bosslevel --envs-per-gpu $n --no-cpu-baseline --steps 512 --warmup 128 2>&1 | line "bosslevel x $n EPW=$epw "
done; done | tee $OUT/bosslevel_lds_record.txt
timeout 200 python bench.py --workload bosslevel --fused 0 --no-cpu-baseline --steps 256 --warmup 64 2>&1 | line "bosslevel x 131072 one launch per step " | tee -a $OUT/bosslevel_lds_record.txt
