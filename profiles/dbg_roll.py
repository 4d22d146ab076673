#!/usr/bin/env python3
"""Debug aid: where does a fused k_roll7 rollout first differ from the oracle?  python profiles/dbg_roll.py ENV N MAX_STEPS NW"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniGrid-DoorKey-8x8-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
ms = int(sys.argv[3]) if len(sys.argv) > 3 else 3
os.environ["MG_ROLL_NW"] = sys.argv[4] if len(sys.argv) > 4 else "1"
import minigrid_amd as mg  # noqa: E402
from oracle import oracle as O  # noqa: E402

env = mg.make_vec(env_id, n, traj_slots=32, max_steps=ms)
orc = O.OracleVec(env_id, n, max_steps=ms)
obs, _ = env.reset(seed=5)
o_obs, _, _ = orc.reset(seeds=np.arange(5, 5 + n, dtype=np.uint64))
print("reset equal:", (obs["image"] == o_obs).all())
bad_total = 0
for c in range(4):
    env.rollout(32, action_seed=9, fused=True)
    for k in reversed(range(32)):
        img, rew, term, trunc, d, m, act = env.trajectory(k)
        oo, orew, oterm, otrunc, od, om = orc.step(act)
        t = c * 32 + 31 - k
        bad = np.argwhere((img != oo).reshape(n, -1).any(1)).ravel()
        flags = (term != oterm).sum() + (trunc != otrunc).sum() + (d != od).sum()
        if bad.size or flags:
            bad_total += 1
            print(f"step {t}: {bad.size} envs differ (first {bad[:8]}), flag mismatches {flags}; wg of first: {bad[:1] // 64}, lane {bad[:1] % 64}")
            if bad_total <= 2:
                e = int(bad[0])
                print(" device type plane:\n", img[e, :, :, 0].T, "\n oracle type plane:\n", oo[e, :, :, 0].T)
                g, a = orc.get_state()
                print(" oracle agent", a[e], "dir dev/orc", d[e], od[e], "act", act[e], "term", term[e], oterm[e], "trunc", trunc[e], otrunc[e])
            if bad_total > 6:
                sys.exit(1)
print("done, bad steps:", bad_total)
