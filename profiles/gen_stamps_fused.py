"""tuning aid: cycle stamps of generator workgroup 0 inside fused k_step launches during a rollout
   (needs build/libmg_debug.so built with -DMG_DEBUG_TIMING)"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MINIGRID_AMD_LIB"] = os.path.join(ROOT, "build", "libmg_debug.so")
import numpy as np
import minigrid_amd as mg
from minigrid_amd import _binding as B
env_id, n = sys.argv[1], int(sys.argv[2])
env = mg.make_vec(env_id, n, obs_mode=sys.argv[3] if len(sys.argv) > 3 else "partial")
L = B.load()
L.mg_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
env.reset(seed=0)
env.rollout(700, action_seed=1)
for k in range(8):
    env.rollout(1, action_seed=2 + k)
    st = np.zeros(12, np.uint64)
    L.mg_debug_stamps(env.handle, st.ctypes.data_as(C.c_void_p))
    d = (st[1:7] - st[0:6]).astype(np.int64)
    print(env_id, "cycles: entry->first env %d, load %d, refill %d, generate %d, final %d, store %d | total %d" % (tuple(d) + (int(st[6] - st[0]),)))
