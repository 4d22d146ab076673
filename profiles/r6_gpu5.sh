#!/bin/bash
# Round 6, call 5: MultiRoom's room-chain search on lanes (mr_spec) + vectorised room walls; place_obj's speculative tries for any range; the cooperative
# spare fetch of k_roll7 (big grids / sentence levels: the wave fetches a resetting env's next grid one step ahead instead of one lane's 31-40 dependent loads).
# GPU suite, the generator attribution after the change, the generator families in both regimes, the pure step rate of the big grids (a window without
# resets: --dephase 0, fewer steps than max_steps), one launch per step.
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call5.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke_call5.log
VAR=$ROOT/minigrid_amd/libminigrid_hip_genattr.so
for spec in "multiroom MiniGrid-MultiRoom-N6-v0 65536 512" "goto BabyAI-GoTo-v0 131072 1024" "bosslevel BabyAI-BossLevel-v0 131072 1024"; do
  set -- $spec
  MINIGRID_AMD_LIB=$VAR timeout 300 python profiles/gen_attr.py $2 $3 $4 > $OUT/refill_attribution_$1_call5.txt 2>&1
  grep -v amdgpu.ids $OUT/refill_attribution_$1_call5.txt
done
for w in babyai_goto bosslevel multiroom keycorridor; do
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 (de-phased)"
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline --dephase 0 2>/dev/null | line "$w steps 1024 --dephase 0"
  python bench.py --workload $w --steps 96 --warmup 0 --no-cpu-baseline --dephase 0 2>/dev/null | line "$w steps 96 --warmup 0 --dephase 0 (no reset in the window)"
  python bench.py --workload $w --fused 0 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | line "$w one launch per step"
done | tee $OUT/bench_lines_generators_call5.txt
for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do
  python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w steps 2048"
done | tee $OUT/bench_lines_baseline_call5.txt
