#!/usr/bin/env python3
"""Where the host time of a driver-sized run (one 5-step and one 20-step fused launch of the headline config) goes: per-call wall clock
of the bracket bench.py uses, 200 repetitions."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import minigrid_amd as mg  # noqa: E402

env = mg.make_vec("MiniGrid-Empty-8x8-v0", 65536, output="torch")
env.reset(seed=0)
env.sync()
pc = time.perf_counter
rows = []
for rep in range(200):
    env.rollout(5, action_seed=1, fused=True)
    env.sync(); torch.cuda.synchronize()
    t = [pc()]
    env.timer_start(); t.append(pc())
    env.rollout(20, action_seed=2, fused=True); t.append(pc())
    ev = env.timer_stop(); t.append(pc())
    env.sync(); t.append(pc())
    torch.cuda.synchronize(); t.append(pc())
    rows.append([(b - a) * 1e6 for a, b in zip(t[:-1], t[1:])] + [ev * 1e3])
r = np.median(np.asarray(rows[20:]), axis=0)
print("median us: timer_start %.1f | enqueue (rollout call) %.1f | timer_stop (record + wait) %.1f | mg_sync %.1f | torch.cuda.synchronize %.1f | event time %.1f"
      % tuple(r))
print("bracket as bench.py counts it (enqueue + wait + mg_sync): %.1f us; minus the event time: %.1f us" % (r[1] + r[2] + r[3], r[1] + r[2] + r[3] - r[5]))
