#!/bin/bash
# round 4: refill geometry by level (sentence levels: 2 generating wavefronts per request segment + ring of 128; big grids: 4): parity, bench lines
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4last2; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 %.3f G %.2f us/step (event %.2f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us']))"; }
timeout 900 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider -k "Boss or sentence or GoToSeq or Synth or OpenTwoDoors or done_actions or MoveTwo or PickupLoc or synths5r2 or wrapping or pickling or MultiRoom or BabyAI" > $OUT/pytest_sentence.log 2>&1; echo "tests rc=$?" | tee $OUT/rc.txt
tail -4 $OUT/pytest_sentence.log
for n in 131072 32768 262144; do timeout 300 python bench.py --workload bosslevel --envs-per-gpu $n --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "bosslevel x $n "; done | tee $OUT/bosslevel_final2.txt
for w in babyai_goto multiroom keycorridor; do timeout 200 python bench.py --workload $w --no-cpu-baseline --steps 1024 --warmup 128 2>&1 | line "$w "; done | tee -a $OUT/bosslevel_final2.txt
timeout 300 python bench.py --workload bosslevel --fused 0 --no-cpu-baseline --steps 256 --warmup 64 2>&1 | line "bosslevel x 131072 one launch per step " | tee -a $OUT/bosslevel_final2.txt
