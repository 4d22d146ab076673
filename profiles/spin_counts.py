"""Tuning aid (attribution build only: python profiles/variant_build.py spin -DMG_ATTRIBUTION -DMG_SPIN_COUNTS, loaded with MINIGRID_AMD_LIB): how long the waves of
k_roll7's log split wait for each other.  Per 32-step launch and workgroup: iterations of the dynamics wave's flow-control loop (it is a full log
ahead of the slowest encode wave) and of the encode waves' wait for the next log entry (they are out of work); one iteration = s_sleep 1 + a poll
(~100 cycles).  Usage: python profiles/spin_counts.py <env id> <n> [launches]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import minigrid_amd as mg
from minigrid_amd import _binding as B
env_id, n = sys.argv[1], int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 16
L = B.load()
assert b"attribution=1" in L.mg_build_info(), "needs the attribution build (MINIGRID_AMD_LIB)"
L.mg_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
env = mg.make_vec(env_id, n)
env.reset(seed=0)
env.rollout(32 * 12, action_seed=1, fused=True)            # (past the plain-store launches of a burst: the nontemporal instantiation)
def read():
    st = np.zeros(12, np.uint64)
    L.mg_debug_stamps(env.handle, st.ctypes.data_as(C.c_void_p))
    return int(st[4]), int(st[5])
d0, e0 = read()
env.timer_start()
env.rollout(32 * launches, action_seed=2, fused=True)
ms = env.timer_stop()
d1, e1 = read()
wgs = (n + 63) // 64
steps = 32 * launches
us = ms * 1e3 / steps
cyc = us * 2400.0                                           # ~2.4 GHz
dyn = (d1 - d0) / wgs / steps
enc = (e1 - e0) / wgs / steps
print(f"{env_id} x {n}: {us:.3f} us/step ({cyc:.0f} cycles) | per workgroup-step: dynamics wave waits {dyn:.2f} iterations (~{100 * dyn / cyc * 100:.0f} % of its time), "
      f"the encode waves together {enc:.2f} iterations (~{100 * enc / 3 / cyc * 100:.0f} % of each one's time)")
