"""bench.py's bookkeeping, without a GPU: the workloads are BASELINE.json's configurations, the algorithmic bytes are SURVEY.md §8(d)'s,
and the committed profiler passes the line quotes (HBM traffic, kernel time) are only used for a run of the same launch shape."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workloads_are_the_baseline_configurations():
    b = _bench()
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    shard = {"empty8x8": (1, 1), "doorkey8x8": (2, 1), "lavacrossing_full": (3, 8), "gotoredball": (4, 8)}   # (config index, GPUs it names)
    for name, (k, gpus) in shard.items():
        env_id, n, obs_mode = b.WORKLOADS[name]
        assert cfgs[k].startswith(env_id), (name, cfgs[k])
        total = int(re.search(r"([0-9][0-9 ]+) envs", cfgs[k]).group(1).replace(" ", ""))
        assert n * gpus == total, (name, n, gpus, total)
        assert (obs_mode == "full") == ("FullyObs" in cfgs[k])


def test_algorithmic_bytes_are_the_surveys():
    b = _bench()
    # 1 + 49*3 + 8 + 8 + 3 + 147 + 8 + 1 + 1 = 324 (partial), + 2 for BabyAI's direction / mission id channel
    assert b.algorithmic_bytes_per_env_step("MiniGrid-Empty-8x8-v0", "partial", 8, 8) == 324
    assert b.algorithmic_bytes_per_env_step("BabyAI-GoToRedBall-v0", "partial", 8, 8) == 326
    assert b.algorithmic_bytes_per_env_step("MiniGrid-LavaCrossingS9N1-v0", "full", 9, 9) == 1 + 81 * 3 + 19 + 81 * 3 + 10


def test_committed_profiles_are_quoted_only_for_the_same_launch_shape():
    b = _bench()
    for name in ("empty8x8", "doorkey8x8", "lavacrossing_full", "gotoredball"):
        env_id, n, obs_mode = b.WORKLOADS[name]
        meta = json.load(open(os.path.join(ROOT, "profiles", "r3", f"meta_{name}.json")))
        assert meta["envs_per_gpu"] == n and meta["steps_per_launch"] == 32
        traffic = b.pmc_traffic_bytes(name, n, 32)
        us = b.rocprof_kernel_us_per_step(name, n, 32)
        W = H = 9 if "Lava" in env_id else 8
        algo = b.algorithmic_bytes_per_env_step(env_id, obs_mode, W, H) * n * 32
        assert traffic is not None and 0.3 * algo < traffic < 1.0 * algo, (name, traffic, algo)     # the grids stay in LDS: about half
        assert us is not None and abs(us - meta["full_launch_avg_us"] / 32) < 1e-9
        # real bytes / kernel time stays below the part's peak
        assert traffic / (us * 32 * 1e-6) < b.HBM_PEAK_GBPS * 1e9
        # a driver-sized run (one 20-step launch) or another batch size must not be given these counters
        assert b.pmc_traffic_bytes(name, n, 20) is None and b.rocprof_kernel_us_per_step(name, n, 20) is None
        assert b.pmc_traffic_bytes(name, n // 2, 32) is None


def test_committed_bench_lines_are_self_consistent():
    """`frac` follows from the line's own fields to 1 %, event time <= host time, steps_per_launch is what ran."""
    for f in ("bench_empty8x8", "bench_doorkey8x8", "bench_lavacrossing_full", "bench_gotoredball", "bench_driver1", "bench_default_run"):
        d = json.loads(open(os.path.join(ROOT, "profiles", "r3", f + ".json")).read().strip().splitlines()[-1])
        r = d["roofline"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "GB/s" and r["bound"] == "hbm"
        assert d["event_ms"] <= d["host_ms"] * 1.001
        assert d["config"]["steps_per_launch"] == min(32, d["steps"])
        assert abs(d["value"] - d["config"]["envs_per_gpu"] * d["steps"] / (d["host_ms"] / 1e3)) / d["value"] < 0.01
        assert abs(d["ms_per_step"] - d["host_ms"] / d["steps"]) / d["ms_per_step"] < 0.01
        n_launch = -(-d["steps"] // d["config"]["steps_per_launch"])
        achieved = r["algorithmic_bytes_per_launch"] / (d["event_ms"] / 1e3 / n_launch) / 1e9
        assert abs(achieved - r["achieved"]) / r["achieved"] < 0.01, (f, achieved, r["achieved"])
        if d["steps"] == 20:
            assert r["traffic"] is None                    # no committed PMC pass has that launch shape
