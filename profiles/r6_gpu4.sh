#!/bin/bash
# Round 6, call 4: the tree after the knob refactor (mg_knobs.h), action staging removed, burst threshold 65 536 for the RoomGrid mazes, k_generate's draw
# buffer at 1024 words: GPU suite, smoke, the generator families in both regimes (de-phased / synchronized), the headline, one launch per step
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
( time timeout 1500 python -m pytest tests -q -m gpu -n 4 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_gpu_call4.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke_call4.log
for w in babyai_goto bosslevel multiroom keycorridor; do
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "$w steps 1024 (de-phased)"
  python bench.py --workload $w --steps 1024 --warmup 128 --no-cpu-baseline --dephase 0 2>/dev/null | line "$w steps 1024 --dephase 0"
done | tee $OUT/bench_lines_generators_call4.txt
python bench.py --workload babyai_goto --steps 2304 --warmup 128 --no-cpu-baseline 2>/dev/null | line "babyai_goto steps 2304 (de-phased)" | tee -a $OUT/bench_lines_generators_call4.txt
for w in empty8x8 doorkey8x8 gotoredball lavacrossing_full; do
  python bench.py --workload $w --steps 2048 --warmup 256 --no-cpu-baseline 2>/dev/null | line "$w steps 2048"
  python bench.py --workload $w --fused 0 --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | line "$w one launch per step"
done | tee $OUT/bench_lines_baseline_call4.txt
for k in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "driver shape run $k"; done | tee -a $OUT/bench_lines_baseline_call4.txt
python - <<'PY' 2>&1 | tail -4 | tee $OUT/reset_latency_call4.txt
import time
import minigrid_amd as mg
for env_id, n in (("BabyAI-GoTo-v0", 131072), ("BabyAI-BossLevel-v0", 131072), ("MiniGrid-MultiRoom-N6-v0", 65536)):
    env = mg.make_vec(env_id, n)
    t0 = time.perf_counter(); env.reset(seed=0); t1 = time.perf_counter(); env.sync(); t2 = time.perf_counter()
    print(f"{env_id} x {n}: reset(seed) returns after {(t1 - t0) * 1e3:.1f} ms, ring (R = {env.spare_ring_depth}) full after {(t2 - t0) * 1e3:.0f} ms")
    env.close()
PY
