"""GPU parity of EXACTLY what bench.py times: `rollout(fused=True)` with the DEFAULT trajectory ring and
spare-episode ring -- 32-step launches, two per refill batch, many launches back to back without a host sync -- at the full
BASELINE.json batch sizes, for at least 2 x max_steps steps, against the CPU oracle replaying the same Philox policy:
every flag of the last 32 steps, observations of sampled steps, the final state and every env's generator position."""
import os

import numpy as np
import pytest

from par_oracle import ParOracle

pytestmark = pytest.mark.gpu

# (env id, envs, FullyObs, max_steps) = BASELINE.json configs[1..4] (per-GPU shards for the 8-GPU configs), as bench.py runs them
CONFIGS = [("MiniGrid-Empty-8x8-v0", 65536, False, 256),
           ("MiniGrid-DoorKey-8x8-v0", 262144, False, 640),
           ("MiniGrid-LavaCrossingS9N1-v0", 131072, True, 324),
           ("BabyAI-GoToRedBall-v0", 32768, False, 64)]


def _check_slot(env, orc_out, slot, what, with_image):
    img, rew, term, trunc, d, m, act = env.trajectory(slot, image=with_image)
    if with_image:
        oo, orew, oterm, otrunc, od, om, oact = orc_out
        assert (img == oo).all(), (what, "image", np.argwhere((img != oo).reshape(len(oo), -1).any(1))[:5].ravel())
        assert (d == od).all() and (m == om).all(), (what, "direction / mission")
    else:
        orew, oterm, otrunc, oact = orc_out
    assert (act == oact).all(), (what, "recorded actions != the oracle's Philox policy")
    assert rew.tobytes() == orew.tobytes(), (what, "reward")
    assert (term == oterm).all() and (trunc == otrunc).all(), (what, "flags")
    return int(term.sum()) + int(trunc.sum())


@pytest.mark.parametrize("env_id,n,full,max_steps", CONFIGS)
def test_bench_configuration_fused_default_rings_full_size(env_id, n, full, max_steps):
    import minigrid_amd as mg
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    env = mg.make_vec(env_id, n, obs_mode="full" if full else "partial")          # default traj_slots / spare_ring, like bench.py
    F = env.max_fused_steps
    assert F == 32 and env.traj_slots == 32, (F, env.traj_slots)
    orc = ParOracle(env_id, n, full)
    obs, _ = env.reset(seed=0)
    assert (obs["image"] == orc.reset(0)[0]).all()
    seed, t = 2, 0
    # ---- phase A: one long call = back-to-back 32-step launches, no host sync in between (what bench.py's timed region does)
    TA = (2 * max_steps + F - 1) // F * F
    if cores < 16:
        TA = (max_steps + 2 * F) // F * F                  # a small host: still past the first truncation burst
    env.rollout(TA, action_seed=seed, fused=True)
    finished = 0
    for j in range(TA - F):
        r, te, tr, _ = orc.philox_step(seed, t); t += 1
        finished += int(te.sum()) + int(tr.sum())
    for k in reversed(range(F)):                           # the ring holds the call's last 32 steps: slot k = k steps before the last
        with_image = k in (F - 1, F // 2 + 1, 0)
        out = orc.philox_step(seed, t, quiet=not with_image); t += 1
        finished += _check_slot(env, out, k, (env_id, "phase A slot", k), with_image)
    assert finished >= n, "every env should have finished at least one episode inside the window"
    g1, a1 = env.get_state(); g2, a2 = orc.get_state()
    assert (g1 == g2).all() and (a1[:, :7] == a2[:, :7]).all()
    assert (env.get_rng_state() == orc.get_rng()).all()
    # ---- phase B: launch by launch, every step's flags; two observations per launch -- and EVERY slot's image in the last launch
    # (the full-size checks sampled observations)
    for c in range(3):
        env.rollout(F, action_seed=seed, fused=True)
        for k in reversed(range(F)):
            with_image = c == 2 or k in (F - 3, 0)
            out = orc.philox_step(seed, t, quiet=not with_image); t += 1
            _check_slot(env, out, k, (env_id, "phase B launch", c, "slot", k), with_image)
    g1, a1 = env.get_state(); g2, a2 = orc.get_state()
    assert (g1 == g2).all() and (a1[:, :7] == a2[:, :7]).all()
    assert (env.get_rng_state() == orc.get_rng()).all()
    assert env.counters()["env_steps"] == n * t
    env.close(); orc.close()


def test_traj_slots_beyond_32_keep_32_step_launches():
    """traj_slots = 64 must not make fused launches longer than the LDS action staging (32 steps)."""
    import minigrid_amd as mg
    import torch
    n, T = 1500, 128
    a_env = mg.make_vec("MiniGrid-DoorKey-8x8-v0", n, traj_slots=64, output="torch")
    b_env = mg.make_vec("MiniGrid-DoorKey-8x8-v0", n)
    assert a_env.max_fused_steps == 32 and a_env.traj_slots == 64
    a_env.reset(seed=3); b_env.reset(seed=3)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 7, (T, n), dtype=np.uint8)
    a_env.step_many(torch.from_numpy(acts).cuda())            # device actions: chunks of max_fused_steps
    a_env.sync()
    for j in range(T):
        obs, rew, term, trunc, _ = b_env.step(acts[j])
        if j >= T - 64:
            img, r2, t2, u2, d2, m2, act = a_env.trajectory(T - 1 - j)
            assert (img == obs["image"]).all() and r2.tobytes() == rew.tobytes() and (t2 == term).all() and (u2 == trunc).all(), j
            assert (act == acts[j]).all()
    ga, aa = a_env.get_state(); gb, ab = b_env.get_state()
    assert (ga == gb).all() and (aa == ab).all()
    a_env.close(); b_env.close()
