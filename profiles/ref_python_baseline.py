#!/usr/bin/env python3
"""Time the UNMODIFIED reference (Farama-Foundation/Minigrid at /root/reference, imported through oracle/gym_shim's
stand-in for the absent gymnasium/pygame packages) with the loop of BASELINE.md §3 = minigrid/benchmark.py:32-43
plumbing, random actions, symbolic observations:

    env = gym.make(id) [+ ImgObsWrapper | FullyObsWrapper];  env.reset(seed=0)
    for a in np.random.default_rng(seed).integers(0, 7, T):  obs, r, term, trunc, _ = env.step(a);  reset() on done

on 1 core and on P = os.cpu_count() independent processes (no IPC: the reference's envs are independent).

/root/reference exists only in the build container, so this runs HERE and its result is committed
(profiles/r2/reference_python_baseline.json); bench.py quotes it next to the oracle-port number it measures live on
the GPU box's host cores.

    python profiles/ref_python_baseline.py [--seconds 8] [--out profiles/r2/reference_python_baseline.json]
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKLOADS = {
    "empty8x8": ("MiniGrid-Empty-8x8-v0", "img"),                 # BASELINE.json configs[0] / [1]
    "doorkey8x8": ("MiniGrid-DoorKey-8x8-v0", "img"),             # configs[2]
    "lavacrossing_full": ("MiniGrid-LavaCrossingS9N1-v0", "full"),  # configs[3]
    "gotoredball": ("BabyAI-GoToRedBall-v0", "dict"),             # configs[4]
}


def _make(env_id, wrap):
    sys.path.insert(0, os.path.join(ROOT, "oracle", "gym_shim"))
    sys.path.insert(0, "/root/reference")
    import gymnasium as gym          # the shim
    import minigrid                  # noqa: F401  (registers the ids)
    from minigrid.wrappers import FullyObsWrapper, ImgObsWrapper
    env = gym.make(env_id)
    if wrap == "img":
        env = ImgObsWrapper(env)
    elif wrap == "full":
        env = FullyObsWrapper(env)
    return env


def run(env_id, wrap, seconds, seed):
    import numpy as np
    env = _make(env_id, wrap)
    env.reset(seed=seed)
    acts = np.random.default_rng(seed).integers(0, 7, 4096)
    n, t0 = 0, time.perf_counter()
    while True:
        for a in acts:
            _, _, term, trunc, _ = env.step(int(a))
            if term or trunc:
                env.reset()
        n += len(acts)
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return n, dt


def _worker(args):
    return run(*args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r2", "reference_python_baseline.json"))
    a = ap.parse_args()
    P = os.cpu_count() or 1
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    import numpy
    out = {"what": "unmodified reference /root/reference (Minigrid v3.1.0) through oracle/gym_shim; loop = minigrid/benchmark.py:32-43 "
                   "plumbing with np.random.default_rng(seed).integers(0, 7) actions, reset() on done (BASELINE.md section 3)",
           "hardware": f"{cpu}, {P} logical cores (build container)", "python": platform.python_version(), "numpy": numpy.__version__,
           "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
           "seconds_per_sample": a.seconds, "workloads": {}}
    for name, (env_id, wrap) in WORKLOADS.items():
        n1, dt1 = run(env_id, wrap, a.seconds, 0)
        with mp.get_context("fork").Pool(P) as pool:
            t0 = time.perf_counter()
            res = pool.map(_worker, [(env_id, wrap, a.seconds, s) for s in range(P)])
            wall = time.perf_counter() - t0
        tot = sum(n for n, _ in res)
        rate_p = sum(n / dt for n, dt in res)
        out["workloads"][name] = {"env_id": env_id, "wrapper": {"img": "ImgObsWrapper", "full": "FullyObsWrapper", "dict": "none (dict obs)"}[wrap],
                                  "one_core": {"value": n1 / dt1, "unit": "env-steps/s", "cores": 1, "steps": n1, "seconds": dt1},
                                  "all_cores": {"value": rate_p, "unit": "env-steps/s", "cores": P, "steps": tot, "wall_seconds": wall}}
        print(name, f"1 core {n1 / dt1:,.0f} steps/s   {P} procs {rate_p:,.0f} steps/s", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
