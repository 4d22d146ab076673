#!/usr/bin/env python3
"""Per-phase cycle attribution of the wave-cooperative episode generators (k_generate / k_refill, mg_genk.h generate_one).

Needs the attribution variant of the library (never the product build):

    python profiles/variant_build.py genattr --units=<cooperative generator units>,mg_api.hip -DMG_GEN_ATTR
    MINIGRID_AMD_LIB=minigrid_amd/libminigrid_hip_genattr.so python profiles/gen_attr.py BabyAI-GoTo-v0 131072 [steps]

Prints, per mode (direct generation of every env = k_generate; the refills of a random-policy rollout = k_refill), the share of each
phase in the generating waves' cycles (s_memtime between marks, summed over every generated episode), cycles per episode and per
whole-level attempt.  MG_LANE_DIRECT=0 / MG_LANE_BURST=0 keep both modes on the cooperative kernels."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MINIGRID_AMD_LIB", os.path.join(ROOT, "minigrid_amd", "libminigrid_hip_genattr.so"))
os.environ.setdefault("MG_LANE_DIRECT", "0")
os.environ.setdefault("MG_LANE_BURST", "0")
import numpy as np

import minigrid_amd as mg
from minigrid_amd import _binding as B

PHASES = ["prologue (stream load, draw refills, restarts)", "room lattice + door offsets", "agent placement", "connect_all",
          "object placement", "reachability flood", "mission / instructions + validation", "epilogue (stream position, stores)",
          "MultiRoom: room-chain search", "MultiRoom: walls + doors"]


def stamps(L, env):
    st = np.zeros(12, np.uint64)
    L.mg_debug_stamps(env.handle, st.ctypes.data_as(C.c_void_p))
    return st.astype(np.int64)


def report(tag, d, wall_s):
    eps, att = int(d[11]), int(d[10])
    tot = int(d[:10].sum())
    if eps == 0:
        print(f"{tag}: no episode generated")
        return
    print(f"{tag}: {eps} episodes, {att / eps:.2f} attempts per episode, {tot / eps:.0f} cycles per episode "
          f"({tot / max(att, 1):.0f} per attempt), wall {wall_s * 1e3:.2f} ms = {eps / wall_s / 1e6:.2f} M episodes/s")
    for k, name in enumerate(PHASES):
        if d[k]:
            print(f"    {100.0 * d[k] / tot:5.1f} %  {d[k] / eps:10.0f} cycles/episode   {name}")


def main():
    env_id, n = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    L = B.load()
    L.mg_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    env = mg.make_vec(env_id, n)
    print(f"# {env_id} x {n}, library {os.environ['MINIGRID_AMD_LIB']}, MG_LANE_DIRECT={os.environ['MG_LANE_DIRECT']} MG_LANE_BURST={os.environ['MG_LANE_BURST']}")
    s0 = stamps(L, env)
    t0 = time.perf_counter()
    env.reset(seed=0)
    env.sync()
    t1 = time.perf_counter()
    s1 = stamps(L, env)
    report("direct generation (reset(seed) + ring fill, k_generate)", s1 - s0, t1 - t0)
    # a random-policy rollout: every episode that ends takes a spare and files a refill request
    env.rollout(64, action_seed=1, fused=True)
    env.sync()
    s2 = stamps(L, env)
    t2 = time.perf_counter()
    left = steps
    while left > 0:
        k = min(left, 32)
        env.rollout(k, action_seed=2, fused=True)
        left -= k
    env.sync()
    t3 = time.perf_counter()
    s3 = stamps(L, env)
    report(f"refills of a {steps}-step random rollout (k_refill)", s3 - s2, t3 - t2)
    print(f"    rollout: {n * steps / (t3 - t2) / 1e9:.2f} G env-steps/s")
    env.close()


if __name__ == "__main__":
    main()
