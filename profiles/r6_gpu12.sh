#!/bin/bash
# Round 6, call 12: MultiRoom on lanes without a grid per lane (mg_genmr.h: 8 KB of LDS per generating wavefront instead of 41): parity, ring fill, refill at several burst thresholds
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('$1 %.3f G env-steps/s  %.3f us/step (event %.3f) episodes in the timed region %d (share %.5f)' % (d['value']/1e9, d['ms_per_step']*1e3, r['avg_step_us'], c.get('episodes_finished_in_timed_region_rank0', -1), c.get('autoreset_share_timed', -1)))
except Exception as ex: print('$1 FAILED', ex)"; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roll.py tests/test_gpu_fused_full.py tests/test_gpu_philox.py -q -m gpu -n 4 -k "MultiRoom or multiroom" 2>&1 | tail -3 | tee $OUT/pytest_gpu_call12.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/reset_latency_call12.txt
import time
import minigrid_amd as mg
for env_id, n in (("MiniGrid-MultiRoom-N6-v0", 65536), ("MiniGrid-MultiRoom-N4-S5-v0", 65536), ("MiniGrid-MultiRoom-N2-S4-v0", 65536)):
    env = mg.make_vec(env_id, n)
    t0 = time.perf_counter(); env.reset(seed=0); t1 = time.perf_counter(); env.sync(); t2 = time.perf_counter()
    R = env.spare_ring_depth
    print(f"{env_id} x {n}: reset(seed) returns after {(t1 - t0) * 1e3:.1f} ms, ring (R = {R}) full after {(t2 - t0) * 1e3:.0f} ms = {n * (R + 1) / (t2 - t0) / 1e6:.1f} M episodes/s")
    env.close()
PY
for cfg in "MG_X=0" "MG_LANE_BURST=0" "MG_LANE_BURST=1024" "MG_LANE_BURST=8192"; do
  env $cfg python bench.py --workload multiroom --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | line "multiroom steps 1024 (de-phased) $cfg"
  env $cfg python bench.py --workload multiroom --steps 1024 --warmup 128 --no-cpu-baseline --dephase 0 2>/dev/null | line "multiroom steps 1024 --dephase 0 $cfg"
done | tee $OUT/bench_lines_multiroom_call12.txt
python bench.py --workload multiroom --fused 0 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | line "multiroom one launch per step" | tee -a $OUT/bench_lines_multiroom_call12.txt
