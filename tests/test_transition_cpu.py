"""The step core on the CPU (no GPU needed): mg_selftest_transition runs env_transition (minigrid_amd/csrc/mg_step.h -- MiniGridEnv.step,
minigrid_env.py:525-595, as the round-4 straight-line form: every effect of every action computed for every lane and selected, one memory access --
plus the level's own step rule) compiled for the host, one step at a time on the oracle's states, against the oracle's own step: grid, agent pose,
carried object, step count, reward bytes, terminated, truncated.  The levels are the ones whose state the exchange format carries completely (no box
contents) and whose rule needs no auxiliary word -- plus the single-room BabyAI GoTo levels (GoToRedBall is BASELINE.json's configs[4]), whose tracked
positions the test builds at every reset and the kernel code keeps current; group / rule / rule_cell / rule_div are what mg_create derives for them (mg_api.hip).  Episodes are
stepped PAST their end now and then (the reference allows it; rewards past max_steps), finished envs reset in batches.
(On the GPU the same code runs inside k_roll7 / k_step against the same oracle and the reference's goldens.)"""
import ctypes as C

import numpy as np
import pytest

from minigrid_amd import _binding as B

GG_NONE, GG_LIGHT, GG_ROOMGRID, GG_ROOMS = 0, 1, 2, 4
RULE_NONE, RULE_FETCH, RULE_GOTODOOR, RULE_UNLOCK, RULE_PICKUP, RULE_REDBLUE, RULE_MEMORY, RULE_PICKUPDESC, RULE_OPENFRONT = 0, 2, 3, 4, 5, 6, 7, 10, 11
T_BALL, T_BOX = 6, 7

# env id -> (kernel variant, rule, rule_cell, rule_div): mg_create's table for these levels
LEVELS = {
    "MiniGrid-Empty-8x8-v0": (GG_NONE, RULE_NONE, 0, 0), "MiniGrid-Empty-Random-6x6-v0": (GG_NONE, RULE_NONE, 0, 0),
    "MiniGrid-DoorKey-8x8-v0": (GG_NONE, RULE_NONE, 0, 0), "MiniGrid-LavaCrossingS9N1-v0": (GG_NONE, RULE_NONE, 0, 0),
    "MiniGrid-SimpleCrossingS9N2-v0": (GG_NONE, RULE_NONE, 0, 0), "MiniGrid-FourRooms-v0": (GG_NONE, RULE_NONE, 0, 0),
    "MiniGrid-LavaGapS7-v0": (GG_NONE, RULE_NONE, 0, 0), "MiniGrid-DistShift1-v0": (GG_NONE, RULE_NONE, 0, 0),
    "MiniGrid-MultiRoom-N2-S4-v0": (GG_NONE, RULE_NONE, 0, 0),
    "MiniGrid-Fetch-8x8-N3-v0": (GG_LIGHT, RULE_FETCH, 0, 0), "MiniGrid-GoToDoor-6x6-v0": (GG_LIGHT, RULE_GOTODOOR, 0, 0),
    "MiniGrid-RedBlueDoors-8x8-v0": (GG_LIGHT, RULE_REDBLUE, 0, 0), "MiniGrid-MemoryS11-v0": (GG_LIGHT, RULE_MEMORY, 0, 0),
    "MiniGrid-Unlock-v0": (GG_ROOMGRID, RULE_UNLOCK, 5, 0), "MiniGrid-UnlockPickup-v0": (GG_ROOMGRID, RULE_PICKUP, T_BOX, 1),
    "MiniGrid-KeyCorridorS3R3-v0": (GG_ROOMGRID, RULE_PICKUP, T_BALL, 1), "MiniGrid-BlockedUnlockPickup-v0": (GG_ROOMGRID, RULE_PICKUP, T_BOX, 2),
    "BabyAI-Pickup-v0": (GG_ROOMS, RULE_PICKUPDESC, 0, 1), "BabyAI-Open-v0": (GG_ROOMS, RULE_OPENFRONT, 0, 6),
    "BabyAI-OpenRedDoor-v0": (GG_ROOMS, RULE_OPENFRONT, 0, 0), "BabyAI-PickupDist-v0": (GG_ROOMS, RULE_PICKUPDESC, 0, 1),
}
POLICY = [0.15, 0.15, 0.35, 0.12, 0.05, 0.13, 0.05]
RULE_GOTO = 1
# the single-room BabyAI GoTo levels (goto.py; BabyAI-GoToRedBall = BASELINE.json configs[4]): GoToInstr succeeds in front of a TRACKED POSITION of a
# described object (verifier.py:309-316; positions refreshed at reset and by drop actions only): the kernel keeps two bitboards per env
GOTO_LEVELS = {"BabyAI-GoToRedBall-v0": (6, 0), "BabyAI-GoToRedBallGrey-v0": (6, 0), "BabyAI-GoToLocal-v0": (0, 2), "BabyAI-GoToObj-v0": (0, 2)}
SORTED_COLOR_TO_IDX = [2, 1, 5, 3, 0, 4]          # blue green grey purple red yellow -> COLOR_TO_IDX


def _described(grid, mission, rule_div):
    """bit y * W + x of every cell holding an object the mission describes (per env)"""
    n, Wd, Ht = grid.shape[:3]
    out = np.zeros(n, np.uint64)
    for i in range(n):
        if rule_div == 0:
            want = (6, 0)                                    # the red ball
        else:
            m18 = int(mission[i]) % 18
            want = (5 + m18 % 3, SORTED_COLOR_TO_IDX[m18 // 3])
        bits = 0
        for x in range(Wd):
            for y in range(Ht):
                if (int(grid[i, x, y, 0]), int(grid[i, x, y, 1])) == want:
                    bits |= 1 << (y * Wd + x)
        out[i] = bits
    return out


def _run(env_id, n, T, no_death=(), death_cost=-1.0):
    from oracle import oracle as O
    L = B.load()
    goto = env_id in GOTO_LEVELS
    group, rule, rule_cell, rule_div = (GG_ROOMGRID, RULE_GOTO) + GOTO_LEVELS[env_id] if goto else LEVELS[env_id]
    kw = dict(no_death_types=no_death, death_cost=death_cost) if no_death else {}
    s = O.spec(env_id)
    Wd, Ht, max_steps = s["width"], s["height"], min(s["max_steps"], 60)       # (short episodes: truncation and steps past it in every run)
    orc = O.OracleVec(env_id, n, max_steps=max_steps, **kw)
    mask = sum(1 << O.OBJECT_TO_IDX[t] for t in no_death)
    orc.reset(seeds=np.arange(50, 50 + n, dtype=np.uint64))
    rng = np.random.default_rng(3)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rew = np.zeros(n, np.float64); term = np.zeros(n, np.uint8); trunc = np.zeros(n, np.uint8); err = np.zeros(n, np.uint32)
    seen = {"term": 0, "trunc": 0, "reward": 0, "carry": 0, "grid": 0}
    done = np.zeros(n, bool)
    aux = np.zeros((n, 2), np.uint64)
    if goto:
        g0, a0 = orc.get_state()
        aux[:, 0] = aux[:, 1] = _described(g0, a0[:, 7], rule_div)
    for t in range(T):
        g0, a0 = orc.get_state()
        grid, agent = g0.copy(), a0.copy()
        act = rng.choice(7, size=n, p=POLICY).astype(np.uint8)
        assert L.mg_selftest_transition(group, rule, rule_cell, rule_div, Wd, Ht, max_steps, mask, float(death_cost), n, p(grid), p(agent), p(act),
                                        p(rew), p(term), p(trunc), p(err), p(aux) if goto else None) == 0
        _, orew, oterm, otrunc, _, _ = orc.step(act, autoreset=0)
        g1, a1 = orc.get_state()
        what = (env_id, t)
        assert not err.any(), what
        bad = np.argwhere((grid != g1).reshape(n, -1).any(1)).ravel()
        assert bad.size == 0, (what, "grid", bad[:4], act[bad[:4]], a0[bad[:4]])
        assert (agent[:, :6] == a1[:, :6]).all(), (what, "agent", np.argwhere((agent[:, :6] != a1[:, :6]).any(1)).ravel()[:4])
        assert rew.tobytes() == orew.tobytes(), (what, "reward", np.argwhere(rew != orew).ravel()[:4])
        assert (term.astype(bool) == oterm).all() and (trunc.astype(bool) == otrunc).all(), (what, "flags")
        seen["term"] += int(oterm.sum()); seen["trunc"] += int(otrunc.sum()); seen["reward"] += int((orew != 0).sum())
        seen["carry"] += int((a1[:, 3] != a0[:, 3]).sum()); seen["grid"] += int((g1 != g0).reshape(n, -1).any(1).sum())
        done |= oterm | otrunc
        # (in between, finished episodes keep being stepped -- except GoToDoor, whose episode ends with a toggle: past its end the agent can walk
        # through the opened door in the outer wall and face the outside of the grid, where the reference asserts)
        if (t % 7 == 6 or "GoToDoor" in env_id) and done.any():
            orc.reset(mask=done.astype(np.uint8))
            if goto:                                         # reset(): the instruction's positions are taken afresh
                g2, a2 = orc.get_state()
                d2 = _described(g2, a2[:, 7], rule_div)
                aux[done, 0] = d2[done]; aux[done, 1] = d2[done]
            done[:] = False
    return seen


@pytest.mark.parametrize("env_id", sorted(LEVELS) + sorted(GOTO_LEVELS))
def test_step_core_on_the_host_equals_the_oracle(env_id):
    seen = _run(env_id, 96, 260 if "Memory" in env_id or "Four" in env_id else 180)
    assert seen["term"] + seen["trunc"] > 0 and seen["grid"] + seen["carry"] >= 0, (env_id, seen)
    if env_id in ("MiniGrid-DoorKey-8x8-v0", "MiniGrid-UnlockPickup-v0", "BabyAI-Pickup-v0", "MiniGrid-Fetch-8x8-N3-v0"):
        assert seen["carry"] > 0 and seen["grid"] > 0, (env_id, seen)


def test_step_core_with_nodeath_on_the_host_equals_the_oracle():
    """NoDeath (wrappers.py:845-882): walking into lava costs death_cost and does not terminate."""
    seen = _run("MiniGrid-LavaCrossingS9N1-v0", 128, 150, no_death=("lava",), death_cost=-0.25)
    assert seen["reward"] > 50, seen
