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
        traffic = b.pmc_traffic_bytes(name, n, 32, any_build=True)
        us = b.rocprof_kernel_us_per_step(name, n, 32, any_build=True)
        W = H = 9 if "Lava" in env_id else 8
        algo = b.algorithmic_bytes_per_env_step(env_id, obs_mode, W, H) * n * 32
        assert traffic is not None and 0.3 * algo < traffic < 1.0 * algo, (name, traffic, algo)     # the grids stay in LDS: about half
        # the latest round's pass of this launch shape is the one quoted
        src = b.pmc_traffic_source(name, n, 32, any_build=True)
        latest = json.load(open(os.path.join(ROOT, os.path.dirname(src), f"meta_{name}.json")))
        assert us is not None and abs(us - latest["full_launch_avg_us"] / 32) < 1e-9
        # real bytes / kernel time stays below the part's peak
        assert traffic / (us * 32 * 1e-6) < b.HBM_PEAK_GBPS * 1e9
        # another launch length or another batch size must not be given these counters
        assert b.pmc_traffic_bytes(name, n, 19, any_build=True) is None and b.rocprof_kernel_us_per_step(name, n, 19, any_build=True) is None
        assert b.pmc_traffic_bytes(name, n // 2, 32, any_build=True) is None


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
            assert r["traffic"] is None                    # (round 3 had no PMC pass of that launch shape; round 4 does: below)


def test_round4_profiles_cover_both_launch_shapes():
    """VERDICT r3 missing #2: the driver's own launch shape (bench.py --steps 20 --warmup 5: one 20-step launch) has committed kernel-trace
    and PMC passes, so its line carries traffic / kernel_us_per_step; the counters agree with the analytic floor to a few per cent and the
    real-bytes fraction stays below 1 for every BASELINE workload."""
    b = _bench()
    r4 = os.path.join(ROOT, "profiles", "r4")
    for name, spl, sfx in (("empty8x8", 32, ""), ("empty8x8", 20, "_spl20"), ("doorkey8x8", 32, ""), ("lavacrossing_full", 32, ""), ("gotoredball", 32, "")):
        env_id, n, obs_mode = b.WORKLOADS[name]
        meta = json.load(open(os.path.join(r4, f"meta_{name}{sfx}.json")))
        assert meta["envs_per_gpu"] == n and meta["steps_per_launch"] == spl
        assert "attribution=0" in meta["library_build"] and not meta["environment"], meta      # the product build, no MG_* switch set
        traffic, us = b.pmc_traffic_bytes(name, n, spl, any_build=True, only_round="r4"), b.rocprof_kernel_us_per_step(name, n, spl, any_build=True, only_round="r4")
        assert b.pmc_traffic_source(name, n, spl, any_build=True, only_round="r4").startswith("profiles/r4/")
        W = H = 9 if "Lava" in env_id else 8
        obe = 3 * W * H if obs_mode == "full" else 147
        floor = (obe + 16 + (2 * W * H + 16) / spl) * n * spl
        assert 0.97 * floor < traffic < 1.12 * floor, (name, spl, traffic / floor)
        frac = traffic / (us * spl * 1e-6) / (b.HBM_PEAK_GBPS * 1e9)
        assert 0.25 < frac < 1.0, (name, frac)
    # round 3's headline fraction on real bytes was 0.54; round 4's committed passes
    t, us = b.pmc_traffic_bytes("empty8x8", 65536, 32, any_build=True, only_round="r4"), b.rocprof_kernel_us_per_step("empty8x8", 65536, 32, any_build=True, only_round="r4")
    assert t / (us * 32e-6) / 8e12 > 0.60


def test_profiles_are_bound_to_the_build_of_the_step_kernels(tmp_path, monkeypatch):
    """a committed PMC / kernel-trace pass is quoted only for the build of the step kernels it measured
    (meta "step_kernel_srchash" == minigrid_amd.build.step_kernel_hash() of this tree); host-only edits (mg_api.hip) do not change the hash."""
    b = _bench()
    from minigrid_amd import build as B
    h = B.step_kernel_hash()
    assert len(h) == 16 and h == b.step_kernel_srchash()
    # rounds 3 and 4 carry no hash: never quoted by a live line any more
    assert b.pmc_traffic_bytes("empty8x8", 65536, 32) is None or \
        any(m.get("step_kernel_srchash") == h for _d, m, _s in b._profile_metas("empty8x8", 65536, 32))
    d = tmp_path / "r9"
    d.mkdir()
    for hh, expect in ((h, True), ("0" * 16, False)):
        (d / "meta_empty8x8.json").write_text(json.dumps({"envs_per_gpu": 65536, "steps_per_launch": 32, "full_launch_avg_us": 64.0,
                                                          "step_kernel_srchash": hh}))
        (d / "pmc_FETCH_SIZE_empty8x8.txt").write_text("FETCH_SIZE,void mg::k_roll7<0, false, true>,calls=20,mean=100.0,total=2000.0,max=120.0\n")
        (d / "pmc_WRITE_SIZE_empty8x8.txt").write_text("WRITE_SIZE,void mg::k_roll7<0, false, true>,calls=20,mean=300000.0,total=1.0,max=340000.0\n")
        monkeypatch.setattr(b, "PROFILES_DIR", str(tmp_path))
        t = b.pmc_traffic_bytes("empty8x8", 65536, 32)
        assert (t == (2 * 120.0 + 340000.0) * 1024) if expect else (t is None)
        assert (b.rocprof_kernel_us_per_step("empty8x8", 65536, 32) == 2.0) if expect else (b.rocprof_kernel_us_per_step("empty8x8", 65536, 32) is None)
    # the hash covers the step units, their headers and the flags -- not the host runtime and not the generators
    units = [u for u in B.UNITS if u.startswith("mg_step_")]
    assert units and "mg_api.hip" not in units and all("mg_gen" not in x for x in units + B._STEP)


def test_round6_profiles_are_quotable_on_this_tree():
    """The committed round-6 passes were taken on THIS build of the step kernels (hash-matched), for both launch shapes of the four BASELINE workloads: a
    live bench line on this tree quotes counters, not the analytic floor; the counters agree with the floor and the real-bytes fraction stays below 1.
    VERDICT r5 "next" #3: the line's frac_profile is what a reader recomputes from profiles/ alone (committed PMC bytes / committed kernel-trace duration),
    the in-run figure is frac_this_run, valu_issue_frac comes from the hash-matched SQ pass, and the driver-shaped window holds finished episodes."""
    b = _bench()
    from minigrid_amd import build as B
    r6 = os.path.join(ROOT, "profiles", "r6")
    for name in ("empty8x8", "doorkey8x8", "lavacrossing_full", "gotoredball"):
        env_id, n, obs_mode = b.WORKLOADS[name]
        for spl, sfx in ((32, ""), (20, "_spl20")):
            meta = json.load(open(os.path.join(r6, f"meta_{name}{sfx}.json")))
            assert meta["step_kernel_srchash"] == B.step_kernel_hash(), (name, sfx, "re-collect profiles/r6 after a change of the step kernels (profiles/r6_final.sh)")
            assert meta["envs_per_gpu"] == n and meta["steps_per_launch"] == spl and meta["full_launches"] >= 15
            assert "attribution=0" in meta["library_build"] and not meta["environment"], meta
            traffic, us = b.pmc_traffic_bytes(name, n, spl), b.rocprof_kernel_us_per_step(name, n, spl)
            assert traffic is not None and us is not None and b.pmc_traffic_source(name, n, spl).startswith("profiles/r6/")
            W = H = 9 if "Lava" in env_id else 8
            obe = 3 * W * H if obs_mode == "full" else 147
            floor = (obe + 16 + (2 * W * H + 16) / spl) * n * spl
            assert 0.97 * floor < traffic < 1.15 * floor, (name, spl, traffic / floor)
            frac = traffic / (us * spl * 1e-6) / (b.HBM_PEAK_GBPS * 1e9)
            assert 0.2 < frac < 1.0, (name, spl, frac)
            valu = b.sq_valu_issue(name, n, spl)
            assert valu is not None and 0.05 < valu[0] < 1.0 and valu[2].startswith("profiles/r6/sq_counters_"), (name, spl, valu)
    # the driver's own line of this round (bench.py --gpus 1 --steps 20 --warmup 5) carries hash-matched traffic, and the three named fractions
    d = json.loads(open(os.path.join(r6, "bench_driver1.json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    assert d["config"]["step_kernel_srchash"] == B.step_kernel_hash() and r["traffic"] is not None
    assert r["traffic_source"].startswith("profiles/r6/") and 0 < r["frac"] <= 1.0 and r["frac"] == r["frac_this_run"]
    n = b.WORKLOADS["empty8x8"][1]
    recomputed = b.pmc_traffic_bytes("empty8x8", n, 20) / (b.rocprof_kernel_us_per_step("empty8x8", n, 20) * 20 * 1e-6) / (b.HBM_PEAK_GBPS * 1e9)
    assert abs(r["frac_profile"] - recomputed) < 1e-9 and 0.2 < r["frac_profile"] < 1.0
    assert abs(r["valu_issue_frac"] - b.sq_valu_issue("empty8x8", n, 20)[0]) < 1e-9
    # the de-phased batch: the 20-step window of the driver's run ends episodes (none did in round 5: 1 of 65 536 envs)
    c = d["config"]
    assert c["dephase"]["groups"] == 32 and c["episodes_finished_in_timed_region_rank0"] > 0
    assert abs(c["autoreset_share_timed"] - c["episodes_finished_in_timed_region_rank0"] / (n * 20)) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_round4_bench_lines_are_self_consistent():
    for f in ("bench_empty8x8", "bench_doorkey8x8", "bench_lavacrossing_full", "bench_gotoredball", "bench_driver1", "bench_default_run"):
        d = json.loads(open(os.path.join(ROOT, "profiles", "r4", f + ".json")).read().strip().splitlines()[-1])
        r = d["roofline"]
        assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["bound"] == "hbm"
        assert r["traffic"] is not None and r["kernel_us_per_step"] is not None and r["traffic_source"].startswith("profiles/r4/")
        assert abs(r["traffic_vs_floor"] - 1) < 0.12
        n_launch = -(-d["steps"] // d["config"]["steps_per_launch"])
        achieved = r["bytes_per_launch"] / (d["event_ms"] / 1e3 / n_launch) / 1e9
        assert abs(achieved - r["achieved"]) / r["achieved"] < 0.01
        assert d["event_ms"] <= d["host_ms"] * 1.001 <= d["host_ms_incl_device_sync"] * 1.002
        assert abs(d["value"] - d["config"]["envs_per_gpu"] * d["steps"] / (d["host_ms"] / 1e3)) / d["value"] < 0.01
        assert d["config"]["library_build"].startswith("attribution=0") and d["config"]["environment"] == {}
        assert r["survey_8d"]["bytes_per_env_step"] in (324, 326, 516)


def _run_bench(args, env_extra=None, timeout=240):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_spawns_n_ranks_by_itself():
    """VERDICT r3 missing #1: `python bench.py --gpus 2` without a launcher environment must start two ranks (here: gloo, a stub env,
    no GPU) and print ONE line whose n_gpus / world size / per-rank clocks say so."""
    r = _run_bench(["--gpus", "2", "--backend", "gloo", "--stub", "--steps", "20", "--warmup", "5"], {"MG_SOME_KNOB": "7"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["distributed"]["world_size"] == 2 and d["config"]["distributed"]["backend"] == "gloo"
    assert len(d["config"]["distributed"]["per_rank_us_per_step"]) == 2
    assert d["config"]["stub"] is True and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["steps_per_launch"] == 20
    assert d["config"]["environment"].get("MG_SOME_KNOB") == "7"          # every MG_* switch of the run is in the line
    assert d["value"] == d["config"]["envs_per_gpu"] * 2 * 20 / (d["host_ms"] / 1e3) or abs(d["value"] * d["host_ms"] / 1e3 / (65536 * 2 * 20) - 1) < 1e-6
    assert 0 < d["roofline"]["frac"] <= 1 and d["host_ms"] <= d["host_ms_incl_device_sync"]


def test_world_size_must_agree_with_gpus():
    """`--gpus 8` inside a 1-rank (or any other) launcher environment is an error, never a silent 1-GPU run."""
    r = _run_bench(["--gpus", "8", "--stub", "--steps", "2", "--warmup", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _run_bench(["--gpus", "1", "--stub", "--steps", "2", "--warmup", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_roofline_fraction_is_priced_on_real_bytes():
    """frac = HBM bytes really moved / kernel time / 8 TB/s.  With the committed counters of a launch shape the line
    quotes them; the section-8(d) figure is a named secondary field."""
    b = _bench()
    n, spl = 65536, 32
    traffic = b.pmc_traffic_bytes("empty8x8", n, spl, any_build=True)
    floor = (147 + 16 + (2 * 64 + 16) / spl) * n * spl
    assert traffic is not None and 0.95 * floor < traffic < 1.10 * floor, (traffic, floor)      # the counters agree with the analytic floor
    assert b.algorithmic_bytes_per_env_step("MiniGrid-Empty-8x8-v0", "partial", 8, 8) * n * spl > 1.8 * traffic   # section 8(d) overcounts ~2x
