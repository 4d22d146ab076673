"""GPU parity for EVERY fused launch length, not only the 16- and 32-step launches the other fused tests use: the
time split of k_roll7 (which wavefront produces which steps of a launch: `split[]`, a function of T) is exercised for
T in {1, 2, 3, 5, 7, 13, 20, 31, 32} -- and for the literal sequence the driver's `bench.py --steps 20 --warmup 5` times, rollout(5) then
rollout(20) -- at the headline batch (65 536 Empty-8x8 envs) and at a ragged 2 085 DoorKey / GoToRedBall envs, EVERY slot of every launch
compared in full (image, reward bytes, flags, direction, mission, recorded action) with the CPU oracle replaying the device's Philox
policy, then the final state and every env's generator position."""
import numpy as np
import pytest

from par_oracle import ParOracle

pytestmark = pytest.mark.gpu

LENGTHS = [1, 2, 3, 5, 7, 13, 20, 31, 32]


def _check_launch(env, orc, seed, t, T, what):
    """The T steps of the launch just made are in slots T-1 .. 0; replay them on the oracle, compare every field of every slot."""
    fin = 0
    for k in reversed(range(T)):
        oo, orew, oterm, otrunc, od, om, oact = orc.philox_step(seed, t, quiet=False); t += 1
        img, rew, term, trunc, d, m, act = env.trajectory(k)
        assert (act == oact).all(), (what, "slot", k, "recorded actions != the oracle's Philox policy")
        assert (img == oo).all(), (what, "slot", k, "image", np.argwhere((img != oo).reshape(len(oo), -1).any(1))[:5].ravel())
        assert rew.tobytes() == orew.tobytes(), (what, "slot", k, "reward")
        assert (term == oterm).all() and (trunc == otrunc).all(), (what, "slot", k, "flags")
        assert (d == od).all() and (m == om).all(), (what, "slot", k, "direction / mission")
        fin += int(term.sum()) + int(trunc.sum())
    return t, fin


def _final_state(env, orc):
    g1, a1 = env.get_state(); g2, a2 = orc.get_state()
    assert (g1 == g2).all() and (a1[:, :7] == a2[:, :7]).all()
    assert (env.get_rng_state() == orc.get_rng()).all()


@pytest.mark.parametrize("env_id,n,max_steps,full", [("MiniGrid-Empty-8x8-v0", 65536, None, False),       # the headline batch as bench.py creates it
                                                      ("MiniGrid-Empty-8x8-v0", 65536, 11, False),         # ... and with resets inside every launch
                                                      ("MiniGrid-DoorKey-8x8-v0", 2085, 9, False), ("BabyAI-GoToRedBall-v0", 2085, 6, False),
                                                      ("MiniGrid-LavaCrossingS9N1-v0", 2085, None, False),
                                                      # FullyObs (k_roll7<., true>: the dynamics wave + encode waves over staged copies of its image-order
                                                      # stream since round 4): resets in nearly every step (lava), a door opened / a key carried, 11 x 11
                                                      ("MiniGrid-LavaCrossingS9N1-v0", 2085, None, True), ("MiniGrid-DoorKey-8x8-v0", 2085, 9, True),
                                                      ("BabyAI-GoToRedBall-v0", 2085, 6, True), ("MiniGrid-LavaCrossingS11N5-v0", 2085, 30, True),
                                                      ("MiniGrid-LavaCrossingS9N1-v0", 131072, None, True)])
def test_every_launch_length_equals_the_oracle(env_id, n, max_steps, full):
    import minigrid_amd as mg
    kw = {} if max_steps is None else {"max_steps": max_steps}
    env = mg.make_vec(env_id, n, obs_mode="full" if full else "partial", **kw)      # default trajectory ring / spare ring, like bench.py
    assert env.max_fused_steps == 32
    orc = ParOracle(env_id, n, full, **kw)
    obs, _ = env.reset(seed=0)
    assert (obs["image"] == orc.reset(0)[0]).all()
    seed, t, fin = 2, 0, 0
    # the driver's run: one 5-step warm-up launch, one 20-step timed launch (bench.py --steps 20 --warmup 5)
    for T in ([5, 20] + LENGTHS + LENGTHS[::-1] if n < 100000 else [5, 20, 32, 32, 7, 32]):
        env.rollout(T, action_seed=seed, fused=True)
        t, f = _check_launch(env, orc, seed, t, T, (env_id, n, "T", T, "t", t))
        fin += f
    if max_steps is not None or "Lava" in env_id:
        assert fin > (n if n < 100000 else n // 2), "episodes should have ended inside the launches"
    _final_state(env, orc)
    assert env.counters()["env_steps"] == n * t
    env.close(); orc.close()


def test_driver_sequence_back_to_back_without_a_host_sync():
    """rollout(5) and rollout(20) enqueued back to back (what the driver's timed region and its warm-up are), read back afterwards: the
    20-step launch's slots hold its steps, slot 0 the last one."""
    import minigrid_amd as mg
    n, seed = 65536, 7
    env = mg.make_vec("MiniGrid-Empty-8x8-v0", n, output="torch")
    orc = ParOracle("MiniGrid-Empty-8x8-v0", n, False)
    env.reset(seed=0); orc.reset(0)
    env.rollout(5, action_seed=seed, fused=True)
    env.rollout(20, action_seed=seed, fused=True)
    env.sync()
    t = 0
    for _ in range(5):
        orc.philox_step(seed, t); t += 1
    t, _ = _check_launch(env, orc, seed, t, 20, "5 + 20")
    _final_state(env, orc)
    env.close(); orc.close()

