"""BabyAI-SynthS5R2-v0 (minigrid/__init__.py:1102-1106): the one id where the reference itself does not always return.

RoomGrid.place_agent (core/roomgrid.py:327-332) retries `while True` until the agent does not face an object.  With 18 objects in six
3 x 3 rooms, about 0.4 % of the episodes leave the agent's room with free cells that ALL face objects in all four directions: the
reference spins for ever.  The device decides that case exactly (mg_gen.h room_stuck) before drawing anything:

* default (`stuck_place_agent="raise"`): identical to the reference wherever the reference comes back (goldens: test_gpu_parity.py runs
  this id's gen / rollout goldens), RecursionError when an env REACHES an episode the reference would hang on -- episodes are drawn
  ahead into the spare ring, so the error is held in the episode record until the episode is taken;
* `stuck_place_agent="redraw"`: the attempt ends like the RecursionError the level's retry loop catches (roomgrid_level.py:58-74) and
  the redrawn map is accepted -- the oracle's restatement, compared here at sizes where the case occurs hundreds of times.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
ID = "BabyAI-SynthS5R2-v0"


def test_reset_raises_where_the_reference_hangs():
    """The (seed, episode) pairs where the unmodified reference never came back (recorded under an alarm, make_golden.py
    main_synths5r2): every earlier reset() of that seed works and equals the oracle, the one that reaches the episode raises."""
    import minigrid_amd as mg
    from oracle import oracle as O
    g = golden(f"gen_{ID}.npz")
    assert len(g["hang_seed"]) >= 3
    for seed, ep_hang in zip(g["hang_seed"], g["hang_episode"]):
        env = mg.make_vec(ID, 1)
        orc = O.OracleVec(ID, 1)
        for ep in range(int(ep_hang)):
            obs, _ = env.reset(seed=int(seed)) if ep == 0 else env.reset()
            o_obs, _, _ = orc.reset(seeds=np.array([seed], np.uint64) if ep == 0 else None)
            assert not orc.stuck().any()
            assert (obs["image"] == o_obs).all() and (obs["mission"] == orc.mission_strings()).all()
        with pytest.raises(RecursionError, match="place_agent"):
            env.reset()
        env.close()


def test_seeded_reset_onto_a_stuck_first_episode_raises():
    """Same through the generator launch that draws the LIVE episode (reset(seed=...)): find seeds whose very first episode is
    stuck with the oracle, check that exactly those raise."""
    import minigrid_amd as mg
    from oracle import oracle as O
    n = 4096
    orc = O.OracleVec(ID, n)
    orc.reset(seeds=np.arange(n, dtype=np.uint64))
    bad = np.flatnonzero(orc.stuck())
    assert 4 <= len(bad) <= 60                       # about 0.4 %
    for s in bad[:3]:
        env = mg.make_vec(ID, 1)
        with pytest.raises(RecursionError, match="place_agent"):
            env.reset(seed=int(s))
        env.close()
    good = np.setdiff1d(np.arange(int(bad[0]) + 3), bad)[:4]
    env = mg.make_vec(ID, len(good))
    obs, _ = env.reset(seed=[int(s) for s in good])
    o2 = O.OracleVec(ID, len(good))
    o_obs, _, _ = o2.reset(seeds=good.astype(np.uint64))
    assert (obs["image"] == o_obs).all()
    env.close()


@pytest.mark.parametrize("fused", [False, True])
def test_autoreset_raises_at_the_step_that_reaches_a_stuck_episode(fused):
    """Stepping: everything equals the oracle until some env's autoreset takes an episode the reference would hang on; that step (the
    launch that contains it) raises."""
    import minigrid_amd as mg
    from oracle import oracle as O
    n = 256
    probe = O.OracleVec(ID, 2 * n)                     # seeds whose FIRST episode the reference comes back from
    probe.reset(seeds=np.arange(5000, 5000 + 2 * n, dtype=np.uint64))
    seeds = (5000 + np.flatnonzero(~probe.stuck())[:n]).astype(np.uint64)
    env = mg.make_vec(ID, n, traj_slots=16)
    orc = O.OracleVec(ID, n)
    obs, _ = env.reset(seed=[int(s) for s in seeds])
    o_obs, _, _ = orc.reset(seeds=seeds)
    assert not orc.stuck().any() and (obs["image"] == o_obs).all()
    rng = np.random.default_rng(1)
    probs = [0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05]
    T, chunk = 4000, 16
    raised_at = None
    for t0 in range(0, T, chunk):
        acts = rng.choice(7, size=(chunk, n), p=probs).astype(np.uint8)
        first_stuck = None
        recs = []
        for j in range(chunk):
            recs.append(orc.step(acts[j]))
            if first_stuck is None and orc.stuck().any():
                first_stuck = j
        try:
            if fused:
                env.step_many(acts)
                if first_stuck is None:
                    for j in (0, chunk - 1):
                        img, rew = env.trajectory(chunk - 1 - j)[:2]
                        assert (img == recs[j][0]).all() and rew.tobytes() == recs[j][1].tobytes(), (t0, j)
                else:
                    env.trajectory(0)
            else:
                for j in range(chunk):
                    obs, rew, term, trunc, _ = env.step(acts[j])
                    assert first_stuck is None or j < first_stuck, "the reference would hang here: RecursionError expected"
                    assert (obs["image"] == recs[j][0]).all() and rew.tobytes() == recs[j][1].tobytes(), (t0, j)
                    assert (term == recs[j][2]).all() and (trunc == recs[j][3]).all()
            assert first_stuck is None, "the reference would hang in this chunk: RecursionError expected"
        except RecursionError:
            assert first_stuck is not None, f"raised in a chunk (t0 = {t0}) where no env reached a stuck episode"
            raised_at = t0
            break
    assert raised_at is not None, "no env reached a stuck episode in this run: lengthen it"
    env.close()


@pytest.mark.parametrize("full", [False, True])
def test_redraw_mode_equals_the_oracle_over_many_episodes(full):
    """stuck_place_agent="redraw" at a size where the case occurs all the time: every observation, reward, flag, mission sentence, the
    final state and every env's stream position equal the oracle's (whose restatement redraws at the same point of the stream)."""
    from test_gpu_parity import _compare_with_oracle
    from oracle import oracle as O
    nterm, ntrunc = _compare_with_oracle(ID, 2048, 400, full, seed0=77, probs=[0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05],
                                         stuck_place_agent="redraw")
    assert nterm + ntrunc > 2048
    # the case was met on the way: these envs draw about 2 048 x 7 episodes, the oracle flags some of the first 3 x 2 048 alone
    orc = O.OracleVec(ID, 2048)
    met = 0
    for ep in range(3):
        orc.reset(seeds=np.arange(77, 77 + 2048, dtype=np.uint64) if ep == 0 else None)
        met += int(orc.stuck().sum())
    assert met >= 2


def test_redraw_mode_fused_rollout_equals_the_oracle():
    """The fused path (k_roll7<GG_SENTENCE>, spare ring, refill stream) on the same level: device-policy rollout against the oracle's
    restatement of the policy, final state + streams."""
    import minigrid_amd as mg
    from oracle import oracle as O
    n, T = 4096, 640
    env = mg.make_vec(ID, n, stuck_place_agent="redraw", rng="pcg64")
    orc = O.OracleVec(ID, n)
    env.reset(seed=9)
    orc.reset(seeds=np.arange(9, 9 + n, dtype=np.uint64))
    env.rollout(T, action_seed=3, fused=True)
    env.sync()
    for t in range(T):
        orc.step(O.philox_actions(3, t, n))
    g1, a1 = env.get_state()
    g2, a2 = orc.get_state()
    assert (g1 == g2).all() and (a1[:, :7] == a2[:, :7]).all()
    assert (env.get_rng_state() == orc.get_rng()).all()
    assert (env.trajectory_missions(0) == orc.mission_strings()).all()
    env.close()


def test_redraw_is_refused_outside_levelgen():
    import minigrid_amd as mg
    with pytest.raises(ValueError):
        mg.make_vec("MiniGrid-DoorKey-8x8-v0", 8, stuck_place_agent="redraw")
    with pytest.raises(ValueError):
        mg.make_vec(ID, 8, stuck_place_agent="skip")
