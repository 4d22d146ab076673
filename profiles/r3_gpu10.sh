#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r3j; mkdir -p $OUT
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %-20s n=%-7d %7.3f G steps/s %6.2f us/step" % (sys.argv[2], d["config"]["env_id"][:20], d["config"]["envs_per_gpu"], d["value"]/1e9, d["ms_per_step"]*1e3))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
echo "== one launch per step: shared encode on / off" | tee $OUT/unfused.txt
for sh in 1 0; do for w in empty8x8 doorkey8x8 gotoredball; do
  MG_ROLL_SHARE=$sh timeout 100 python bench.py --workload $w --fused 0 --steps 512 --warmup 64 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err; line $OUT/b.json ${w}_unfused_share$sh
done; done 2>&1 | tee -a $OUT/unfused.txt
echo "== host overhead of a driver-sized run"
python - <<'PY' 2>&1 | tee $OUT/host_overhead.txt
import time, sys
sys.path.insert(0, ".")
import torch
import minigrid_amd as mg
env = mg.make_vec("MiniGrid-Empty-8x8-v0", 65536, output="torch")
env.reset(seed=0); env.sync(); env.rollout(5, action_seed=1, fused=True); env.sync(); torch.cuda.synchronize()
for rep in range(3):
    t = [time.perf_counter()]
    env.timer_start(); t.append(time.perf_counter())
    env.rollout(20, action_seed=2, fused=True); t.append(time.perf_counter())
    ev = env.timer_stop(); t.append(time.perf_counter())
    env.sync(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["timer_start", "rollout (enqueue)", "timer_stop (waits for the launch)", "sync", "torch.cuda.synchronize"]
    print("  ".join(f"{n} {1e6*(b-a):.1f} us" for n, a, b in zip(names, t, t[1:])), f"| total {1e6*(t[-1]-t[0]):.1f} us, event {1e3*ev:.1f} us")
env.close()
PY
echo "== full GPU suite (one-step launches now share the encode)"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_full.log
