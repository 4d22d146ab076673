"""The REAL kernels on the CPU (no GPU needed): tests/emu compiles the product's HIP sources (minigrid_amd/csrc/*.hip, unchanged) as plain C++
against a stand-in <hip/hip_runtime.h> and runs them on a host SIMT emulator -- one fiber per lane, cross-lane operations (__ballot, __shfl,
readfirstlane, the sources' LDS hand-off markers, __syncthreads, s_sleep) resolved per wavefront, LDS an array, streams in enqueue order
(tests/emu/emu_runtime.cpp).  Through the ordinary Python facade and C ABI (MINIGRID_AMD_LIB -> the emulated library, in a subprocess) this
runs mg_create / reset / fused rollouts / single steps: the generator kernels and their ring protocol, every loop shape of k_roll7 (one wave, the
time split, the LOG split with its LDS step log, the STAGED split of DynamicObstacles / FullyObs / the sentence levels, the shared encode of
one-step launches), NEXT_STEP and in-kernel SAME_STEP autoreset -- against the oracle: every slot's image, reward bytes, flags, direction,
mission, then the final state and every env's stream position.  What the per-function host selftests (test_abi_cpu / test_transition_cpu /
test_verifier_cpu / test_generators_cpu) cannot see -- the wave-level plumbing -- is what this file adds on the CPU; the GPU suite stays the
parity gate (timing, memory ordering and register limits are not modelled).

The second half runs the MG_LANE_WIDE variant of the generator kernels (mg_genlane.h: one lane per episode for EVERY level, written after
round 4's GPU minutes were spent): its kernels have not run on a GPU yet -- here they do run, for 30 levels of every kernel group in depth and for every
registered id once."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

PRODUCT_CASES = [
    # the four BASELINE.json levels: fused launches of the driver's lengths, every split width, resets in nearly every step, then single steps
    {"env": "MiniGrid-Empty-8x8-v0", "n": 200, "launches": [5, 20, 32], "max_steps": 11, "stepped": 5},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 130, "launches": [32, 7], "max_steps": 4, "knobs": {"MG_ROLL_NW": "4"}, "stepped": 3},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 130, "launches": [32, 7], "max_steps": 4, "knobs": {"MG_ROLL_NW": "3"}},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 130, "launches": [32, 7], "max_steps": 4, "knobs": {"MG_ROLL_NW": "2"}},
    {"env": "MiniGrid-Empty-8x8-v0", "n": 100, "launches": [32], "max_steps": 5, "knobs": {"MG_ROLL_NW": "1"}},
    {"env": "MiniGrid-Empty-8x8-v0", "n": 100, "launches": [32], "max_steps": 5, "knobs": {"MG_ROLL_SPLIT": "0", "MG_ROLL_NW": "4"}},
    {"env": "BabyAI-GoToRedBall-v0", "n": 100, "launches": [32, 13], "max_steps": 3},
    {"env": "MiniGrid-LavaCrossingS9N1-v0", "n": 100, "launches": [32, 5, 20], "full": True, "stepped": 3},
    {"env": "MiniGrid-LavaCrossingS9N1-v0", "n": 100, "launches": [32, 5], "full": True, "knobs": {"MG_ROLL_SPLIT": "0"}},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 100, "launches": [32, 5], "full": True, "max_steps": 9, "stepped": 3},
    # DynamicObstacles inside the fused kernel, SAME_STEP autoreset inside the kernels
    {"env": "MiniGrid-Dynamic-Obstacles-6x6-v0", "n": 100, "launches": [5, 20, 3], "stepped": 3},
    {"env": "MiniGrid-Dynamic-Obstacles-Random-6x6-v0", "n": 70, "launches": [20, 3], "autoreset": "same_step"},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 70, "launches": [20, 3], "max_steps": 5, "autoreset": "same_step", "stepped": 4},
    # other single-room levels of the lane generators
    {"env": "MiniGrid-FourRooms-v0", "n": 70, "launches": [32, 13], "max_steps": 20},
    {"env": "MiniGrid-Fetch-8x8-N3-v0", "n": 70, "launches": [32, 13], "max_steps": 20, "stepped": 3},
    {"env": "BabyAI-GoToLocal-v0", "n": 70, "launches": [32, 13], "max_steps": 10},
    # round 5: the Unlock family and KeyCorridor on the per-function lane kernels (k_refill_lane<R, 2> / <R, 5>) in the PRODUCT build
    {"env": "MiniGrid-KeyCorridorS3R3-v0", "n": 100, "launches": [32, 7], "max_steps": 12, "stepped": 3},
    {"env": "MiniGrid-KeyCorridorS3R3-v0", "n": 6, "launches": [16, 16], "max_steps": 6, "spare_ring": 4},
    {"env": "BabyAI-KeyCorridorS4R3-v0", "n": 40, "launches": [32], "max_steps": 10},
    {"env": "MiniGrid-Unlock-v0", "n": 70, "launches": [32, 13], "max_steps": 10},
    {"env": "MiniGrid-UnlockPickup-v0", "n": 70, "launches": [32], "max_steps": 10, "stepped": 3},
    {"env": "MiniGrid-BlockedUnlockPickup-v0", "n": 70, "launches": [32], "max_steps": 10, "autoreset": "same_step"},
    # round 5: the burst hybrid of the wavefront-per-episode levels -- a batch of at least MG_LANE_BURST requests (default 32 768: a synchronized truncation
    # burst) refills on packed lanes (k_seg_scan + k_refill_lane_packed), a smaller one on k_refill; thresholds that make both halves run here
    {"env": "BabyAI-GoTo-v0", "n": 200, "launches": [32, 32, 7], "max_steps": 10, "spare_ring": 8, "knobs": {"MG_LANE_BURST": "150", "MG_LANE_DIRECT": "1"}},
    {"env": "MiniGrid-MultiRoom-N6-v0", "n": 300, "launches": [32, 32, 7], "max_steps": 10, "spare_ring": 8, "knobs": {"MG_LANE_BURST": "200", "MG_LANE_DIRECT": "1"}},
    {"env": "BabyAI-GoToSeqS5R2-v0", "n": 130, "launches": [32, 32, 32], "spare_ring": 8, "knobs": {"MG_LANE_BURST": "2", "MG_LANE_LPW": "7", "MG_LANE_DIRECT": "1"}},
    {"env": "BabyAI-PutNextS5N2Carrying-v0", "n": 100, "launches": [32, 16], "max_steps": 4, "spare_ring": 8, "knobs": {"MG_LANE_BURST": "60", "MG_LANE_DIRECT": "1"}},
    # ... and the packed refill for the levels whose refill runs on lanes (MG_LANE_PACKED=1: A/B)
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 300, "launches": [32, 32, 7], "max_steps": 4, "knobs": {"MG_LANE_PACKED": "1"}},
    {"env": "BabyAI-GoToRedBall-v0", "n": 200, "launches": [32, 13], "max_steps": 3, "knobs": {"MG_LANE_PACKED": "1", "MG_LANE_LPW": "16"}},
    # round 6: lane refills draw at most MG_LANE_CAP ring slots per request while the ring is half full (GenArgs::slot_cap; rings of 16: the cap is live);
    # MultiRoom's grid-free lane kernels (mg_genmr.h) with every refill on packed lanes
    {"env": "BabyAI-GoToRedBall-v0", "n": 100, "launches": [32, 32, 13, 32], "max_steps": 3, "spare_ring": 16, "knobs": {"MG_LANE_CAP": "1"}},
    {"env": "MiniGrid-KeyCorridorS3R3-v0", "n": 100, "launches": [32, 7, 32], "max_steps": 5, "spare_ring": 16, "knobs": {"MG_LANE_CAP": "2"}},
    {"env": "MiniGrid-MultiRoom-N6-v0", "n": 200, "launches": [32, 32, 7, 32], "max_steps": 5, "spare_ring": 16, "knobs": {"MG_LANE_BURST": "1", "MG_LANE_DIRECT": "1", "MG_LANE_CAP": "2"}},
    # k_step: the other observation modes (one-hot, symbolic, ViewSizeWrapper, FullyObs above 341 cells), DynamicObstacles' round-3 launches
    # (live refill + k_move_obstacles) under FullyObs, RGB frames (tile map + k_render)
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 70, "launches": [16], "max_steps": 6, "obs_mode": "onehot", "stepped": 3},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 70, "launches": [16], "max_steps": 6, "obs_mode": "symbolic", "stepped": 3},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 70, "launches": [16], "max_steps": 6, "view": 5, "stepped": 3},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 70, "launches": [16], "max_steps": 6, "view": 9},
    {"env": "MiniGrid-FourRooms-v0", "n": 40, "launches": [16], "max_steps": 20, "full": True, "stepped": 3},
    {"env": "MiniGrid-Dynamic-Obstacles-6x6-v0", "n": 70, "launches": [8], "full": True, "stepped": 4},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 20, "launches": [], "max_steps": 6, "obs_mode": "rgb", "stepped": 8},
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 20, "launches": [], "max_steps": 6, "obs_mode": "rgb_partial", "stepped": 8},
    # the wavefront-per-episode generators (k_generate / k_refill: WavePcg64's jumped-ahead draws, ballots over the cells, the draw-budget restarts)
    # of every generator group, a handful of envs with a ring of four so that refills happen; the sentence levels' k_roll7 with the verifier
    {"env": "MiniGrid-RedBlueDoors-8x8-v0", "n": 6, "launches": [16, 16], "max_steps": 6, "spare_ring": 4},
    {"env": "MiniGrid-MemoryS7-v0", "n": 6, "launches": [16, 16], "max_steps": 4, "spare_ring": 4},
    {"env": "MiniGrid-MultiRoom-N4-S5-v0", "n": 6, "launches": [16, 16], "max_steps": 6, "spare_ring": 4},
    {"env": "BabyAI-PutNextS5N2Carrying-v0", "n": 6, "launches": [16, 16], "max_steps": 4, "spare_ring": 4},
    {"env": "BabyAI-GoToObjMaze-v0", "n": 4, "launches": [16], "max_steps": 6, "spare_ring": 4},
    {"env": "BabyAI-MiniBossLevel-v0", "n": 4, "launches": [16], "spare_ring": 4, "stepped": 3},
    {"env": "BabyAI-BossLevel-v0", "n": 4, "launches": [16], "spare_ring": 4, "autoreset": "same_step"},
]
_W = lambda env, n=40, **kw: dict({"env": env, "n": n, "launches": [32], "max_steps": 10}, **kw)
WIDE_CASES = [
    {"env": "MiniGrid-KeyCorridorS3R3-v0", "n": 100, "launches": [32, 7], "max_steps": 12, "stepped": 3},
    {"env": "BabyAI-BossLevel-v0", "n": 70, "launches": [32, 7], "stepped": 3},
    {"env": "BabyAI-BossLevel-v0", "n": 70, "launches": [20, 3], "autoreset": "same_step", "stepped": 3},
    {"env": "BabyAI-GoToSeqS5R2-v0", "n": 70, "launches": [32, 32, 32, 32]},            # (episodes end by success / their own step limit: refills)
    _W("BabyAI-SynthS5R2-v0", 70), _W("BabyAI-MiniBossLevel-v0", 70),
    _W("BabyAI-MoveTwoAcrossS5N2-v0", 70, max_steps=8), _W("BabyAI-OpenTwoDoors-v0", 70, max_steps=8), _W("BabyAI-OpenDoorsOrderN4-v0", 70, max_steps=8),
    {"env": "MiniGrid-MultiRoom-N6-v0", "n": 70, "launches": [32, 7], "max_steps": 10},
    _W("MiniGrid-MemoryS7-v0", 70, max_steps=5), _W("BabyAI-PutNextS5N2Carrying-v0", 70, max_steps=4), _W("MiniGrid-UnlockPickup-v0", 70),
    _W("MiniGrid-BlockedUnlockPickup-v0", 70), _W("MiniGrid-GoToDoor-8x8-v0", 70, max_steps=6), _W("MiniGrid-RedBlueDoors-8x8-v0", 70),
    _W("MiniGrid-LockedRoom-v0"), _W("MiniGrid-Playground-v0"), _W("MiniGrid-ObstructedMaze-Full-v1"), _W("MiniGrid-PutNear-8x8-N3-v0", 70, max_steps=5),
    _W("BabyAI-GoTo-v0"), _W("BabyAI-Pickup-v0"), _W("BabyAI-UnblockPickup-v0"), _W("BabyAI-KeyInBox-v0"), _W("BabyAI-PutNextS7N4-v0"),
    _W("BabyAI-ActionObjDoor-v0"), _W("BabyAI-FindObjS5-v0"), _W("BabyAI-UnlockLocal-v0"), _W("BabyAI-PickupDist-v0"), _W("BabyAI-OpenRedDoor-v0"),
    _W("BabyAI-KeyCorridorS4R3-v0"),
    # the single-room levels keep the product kernel (FN = 0) in the wide build
    {"env": "MiniGrid-DoorKey-8x8-v0", "n": 70, "launches": [32], "max_steps": 4},
]


def _run(defines, cases, **extra_env):
    import build_emu
    lib = build_emu.build(defines)
    env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1")
    for k in [k for k in env if k.startswith("MG_") or k.startswith("EMU_")]:
        del env[k]
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_cases.py"), json.dumps(cases)], env=env, capture_output=True, text=True,
                         timeout=900)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == len(cases), (out.returncode, out.stdout[-2000:], out.stderr[-3000:])
    bad = [(r["case"], r.get("error"), r.get("where")) for r in lines if not r["ok"]]
    assert not bad, bad
    if "EMU_SCHED_SEED" in extra_env:
        assert f"emu: seeded schedule {int(extra_env['EMU_SCHED_SEED'])}" in out.stderr
    return lines


def test_product_kernels_on_the_emulator_equal_the_oracle():
    lines = _run([], PRODUCT_CASES)
    assert sum(r["episodes"] for r in lines) > 5000         # the cases are reset-heavy on purpose: spares taken, rings refilled


def test_lane_wide_variant_on_the_emulator_equals_the_oracle():
    lines = _run([], WIDE_CASES, MG_LANE_BURST="1", MG_LANE_DIRECT="1")
    assert sum(r["episodes"] for r in lines) > 4000


def all_id_cases():
    """One small case per registered id (BabyAI-SynthS5R2-v0 aside: its reference resets hang on some seeds, tests/test_gpu_synths5r2.py): 40 envs, fused
    launches of 16 and 5 steps, 2 single steps, a ring of 4 spares, episodes ending every few steps (the sentence levels keep their own step limit)."""
    from conftest import ALL_IDS, SENTENCE_IDS, STUCK_IDS
    cases = []
    for i in ALL_IDS:
        if i in STUCK_IDS:
            continue
        c = {"env": i, "n": 40, "launches": [16, 5], "spare_ring": 4, "stepped": 2}
        if i not in SENTENCE_IDS:
            c["max_steps"] = 6
        cases.append(c)
    return cases


def test_every_id_on_the_lane_wide_variant():
    """All 171 ids with EVERY refill on packed lanes (MG_LANE_BURST=1: the burst half of the hybrid refill takes every batch; since round 5 every lane
    kernel is in the product build -- direct generation runs on lanes for every level anyway): 64 episodes per emulated wavefront, which is why this takes
    a minute -- the wavefront-per-episode generators take two for the same list (profiles/emu_all_ids.py, profiles/r4/emu_all_ids*.txt)."""
    cases = all_id_cases()
    assert len(cases) >= 170
    _run([], cases, MG_LANE_BURST="1", MG_LANE_DIRECT="1")


def test_product_kernels_under_other_legal_schedules():
    """The emulator's default schedule is one of many the device may take.  EMU_SCHED_SEED shuffles what is free: the order of a grid's workgroups, whose
    turn it is among the waves of a workgroup, where a wave is preempted (after any cross-lane operation), ascending or descending lanes.  The LOG /
    STAGED split rings of k_roll7 (a dynamics wave ahead of its encode waves by up to the ring depth, or starved by them), the shared encode, the
    generator rings: same parity under every seed."""
    for seed in ("1", "2"):
        _run([], PRODUCT_CASES, EMU_SCHED_SEED=seed)
    _run([], WIDE_CASES, EMU_SCHED_SEED="4", MG_LANE_BURST="1", MG_LANE_DIRECT="1")


def test_results_do_not_depend_on_uninitialised_memory():
    """The emulator fills what nobody has written yet with patterns -- device buffers 0xA5, pinned host memory 0xBE, the LDS at workgroup start 0xCD
    -- and EMU_FILL xors them: 0xFF = the complements, 0xA5 = device memory that happens to be ZERO (a fresh box).  The same parity under every fill
    means no kernel's result depends on memory it (or the host) did not write first.  (Uninitialised REGISTERS / locals were checked once with
    -ftrivial-auto-var-init=pattern)"""
    for fill in ("0xFF", "0xA5"):
        _run([], PRODUCT_CASES, EMU_FILL=fill)
    _run([], WIDE_CASES, EMU_FILL="0xFF", MG_LANE_BURST="1", MG_LANE_DIRECT="1")


def test_the_product_never_loads_the_emulator():
    """The emulated library is test infrastructure: it says so in mg_build_info(), and bench.py refuses it."""
    import build_emu
    lib = build_emu.build([])
    env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode != 0 and "emulator" in (out.stderr + out.stdout)
    from minigrid_amd import _binding as B
    assert b"emulator=1" not in B.load().mg_build_info()
