"""GPU stress test of the inter-wave LDS protocol of k_roll7's split loops.

The dynamics wave of a workgroup publishes a step (a log entry, or a staged code block) with two DS writes -- data, then a counter -- separated only by a
compiler barrier, and the encode waves poll the counter: the protocol rests on "the DS operations of one wave execute in order" (mg_roll.h MG_WAVE_ORDER).
The host emulator MODELS that property; only hardware can falsify it.  So: full-size batches (every CU holds its four workgroups, the SIMDs are shared
by dynamics and encode waves of different workgroups), many back-to-back launches, and EVERY byte of EVERY step record of every launch compared with the
CPU oracle -- a log entry read before it was written, or a staging overwritten before its last read, shows up as a wrong observation somewhere.

The same file runs against a stress build of the library (profiles/r5_protocol_stress.sh: -DMG_ROLL_LOG_STEPS=2, MG_DRING=2): with a two-entry log the
dynamics wave runs into its flow control in nearly every step, i.e. both directions of the protocol are exercised ~32 times per launch and workgroup."""
import numpy as np
import pytest

from par_oracle import ParOracle

pytestmark = pytest.mark.gpu

# (env id, envs, FullyObs): the log split (Empty, DoorKey, GoToRedBall), the staged split (FullyObs; DynamicObstacles; a sentence level)
CASES = [("MiniGrid-Empty-8x8-v0", 65536, False),
         ("MiniGrid-DoorKey-8x8-v0", 65536, False),
         ("BabyAI-GoToRedBall-v0", 32768, False),
         ("MiniGrid-LavaCrossingS9N1-v0", 65536, True),
         ("MiniGrid-Dynamic-Obstacles-8x8-v0", 32768, False),
         ("BabyAI-GoToLocalS8N7-v0", 16384, False)]
LAUNCHES = 6


@pytest.mark.parametrize("env_id,n,full", CASES)
def test_every_record_of_back_to_back_full_size_launches(env_id, n, full):
    import minigrid_amd as mg
    env = mg.make_vec(env_id, n, obs_mode="full" if full else "partial")
    F = env.max_fused_steps
    assert F == 32 and env.traj_slots >= F
    orc = ParOracle(env_id, n, full)
    obs, _ = env.reset(seed=11)
    assert (obs["image"] == orc.reset(11)[0]).all()
    seed, t = 5, 0
    for c in range(LAUNCHES):
        if c == 2:
            # two launches enqueued back to back without a host sync in between (the first one's records are overwritten: replayed quietly)
            env.rollout(2 * F, action_seed=seed, fused=True)
            for _ in range(F):
                orc.philox_step(seed, t); t += 1
        else:
            env.rollout(F, action_seed=seed, fused=True)
        for k in reversed(range(F)):
            img, rew, term, trunc, d, m, act = env.trajectory(k)
            oo, orew, oterm, otrunc, od, om, oact = orc.philox_step(seed, t, quiet=False); t += 1
            what = (env_id, "launch", c, "slot", k)
            assert (act == oact).all(), what
            bad = (img != oo).reshape(n, -1).any(1)
            assert not bad.any(), (what, "image: envs", np.flatnonzero(bad)[:8], "workgroups", np.unique(np.flatnonzero(bad) // 64)[:8])
            assert rew.tobytes() == orew.tobytes() and (term == oterm).all() and (trunc == otrunc).all(), (what, "scalars")
            assert (d == od).all() and (m == om).all(), (what, "direction / mission")
    g1, a1 = env.get_state(); g2, a2 = orc.get_state()
    assert (g1 == g2).all() and (a1[:, :7] == a2[:, :7]).all()
    assert (env.get_rng_state() == orc.get_rng()).all()
    env.close(); orc.close()
