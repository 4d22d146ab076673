"""The sentence levels' verifier on the CPU (no GPU needed): mg_selftest_verify runs the product's verify_action (minigrid_amd/csrc/mg_verify.h --
RoomGridLevel.step's instrs.verify + object identity, reorganised for SIMT execution in round 4: one scan of the position table per step, the
leaves' results computed up front, side effects applied to the leaves the tree walk looked at, the record's hot words in registers) compiled for
the host; tests/verify_ref.py is a literal, sequential restatement of ActionInstr.verify / And / Before / After (verifier.py:228-571) and of the
identity bookkeeping, leaf by leaf and call by call.  Random records (every tree shape of LevelGen's grammar, every verb, strict modes,
use_done_actions, full stale lists) over random states built so that they are REACHABLE -- objects on distinct cells, the carried object
nowhere on the grid, an action's effect (pickup, drop, an opened box) visible in front of the agent -- must give identical statuses, step
limits, error bits and records.  (On the GPU the same code is compared with the oracle and the reference's goldens: tests/test_gpu_parity.py,
tests/test_gpu_roll.py.)"""
import ctypes as C

import numpy as np
import pytest

from minigrid_amd import _binding as B

import verify_ref as V

W = H = 22
T_KEY, T_BALL = 5, 6


def _case(rng, done_actions):
    """One reachable (state after the action, action, record before the verifier ran) triple."""
    grid = np.zeros((W, H, 3), np.uint8)
    grid[:, :, 0] = V.T_EMPTY
    grid[0, :, 0] = grid[-1, :, 0] = grid[:, 0, 0] = grid[:, -1, 0] = V.T_WALL
    grid[0, :, 1] = grid[-1, :, 1] = grid[:, 0, 1] = grid[:, -1, 1] = 5
    K = int(rng.integers(3, 24))
    cells = rng.choice((W - 2) * (H - 2), K + 1, replace=False)
    xy = [(1 + int(c) % (W - 2), 1 + int(c) // (W - 2)) for c in cells]
    objs = []
    for i in range(K):
        t = int(rng.choice([V.T_DOOR, T_KEY, T_BALL, V.T_BOX], p=[0.2, 0.3, 0.25, 0.25]))
        objs.append((t, int(rng.integers(0, 6)), int(rng.integers(0, 3)) if t == V.T_DOOR else 0))
    pos = [V.POS_GONE] * 64
    for i, (x, y) in enumerate(xy[:K]):
        grid[x, y] = objs[i]
        pos[i] = y * W + x
    # the agent: next to an object more often than chance would have it
    scenario = rng.choice(["none", "carrying", "picked", "dropped", "box_gone", "face"], p=[0.2, 0.15, 0.2, 0.2, 0.1, 0.15])
    d = int(rng.integers(0, 4))
    target = int(rng.integers(0, K))
    if scenario in ("picked", "dropped", "box_gone", "face"):
        tx, ty = xy[target]
        ax, ay = tx - V.DX[d], ty - V.DY[d]
        if not (1 <= ax < W - 1 and 1 <= ay < H - 1) or grid[ax, ay, 0] != V.T_EMPTY:
            scenario = "none"
    if scenario in ("none", "carrying"):
        ax, ay = xy[K]
    carried_hdr, carry_t, carry_c = 0, 0, 0
    act = int(rng.integers(0, 7))
    pickable = [i for i in range(K) if objs[i][0] != V.T_DOOR]
    if scenario == "carrying" and pickable:                 # something in hand since an earlier step
        c = int(rng.choice(pickable))
        grid[xy[c]] = (V.T_EMPTY, 0, 0); pos[c] = V.POS_CARRIED
        carried_hdr, carry_t, carry_c = c + 1, objs[c][0], objs[c][1]
        if act in (V.A_PICKUP, V.A_DROP) and rng.random() < 0.5:
            act = int(rng.choice([0, 1, 2, 5, 6]))
    elif scenario == "picked" and objs[target][0] != V.T_DOOR:   # this step's pickup: the record has not seen it yet
        grid[xy[target]] = (V.T_EMPTY, 0, 0)
        carry_t, carry_c = objs[target][0], objs[target][1]
        act = V.A_PICKUP
        if rng.random() < 0.03:
            pos[target] = V.POS_GONE                             # (an object the record does not track: the tracking error, not a crash)
    elif scenario == "dropped" and objs[target][0] != V.T_DOOR:  # this step's drop: the object lies in front, the record still says "carried"
        pos[target] = V.POS_CARRIED
        carried_hdr = target + 1
        act = V.A_DROP
    elif scenario == "box_gone" and objs[target][0] == V.T_BOX:  # this step's toggle opened a box: Box.toggle put its (empty) content there
        grid[xy[target]] = (V.T_EMPTY, 0, 0)
        act = V.A_TOGGLE
    elif scenario == "face" and objs[target][0] == V.T_DOOR and rng.random() < 0.6:
        act = V.A_TOGGLE
    words = [0] * 40
    # the tree: a leaf | And(l, l) | Before / After over leaves or And nodes, distinct leaves (levelgen.py:157-211)
    leaves = [int(v) for v in rng.permutation(4)]
    shape = int(rng.integers(0, 6))
    nodes = [0, 0, 0]
    mk = lambda kind, a_, b_: kind | (a_ << 2) | (b_ << 5)
    if shape == 0:
        root = leaves[0]
    elif shape == 1:
        root, nodes[0] = 4, mk(V.N_AND, leaves[0], leaves[1])
    else:
        kind = int(rng.choice([V.N_BEFORE, V.N_AFTER]))
        if shape == 2:
            root, nodes[0] = 4, mk(kind, leaves[0], leaves[1])
        elif shape == 3:
            root, nodes[0], nodes[1] = 4, mk(kind, 5, leaves[2]), mk(V.N_AND, leaves[0], leaves[1])
        elif shape == 4:
            root, nodes[0], nodes[1] = 4, mk(kind, leaves[2], 5), mk(V.N_AND, leaves[0], leaves[1])
        else:
            root, nodes[0], nodes[1], nodes[2] = 4, mk(kind, 5, 6), mk(V.N_AND, leaves[0], leaves[1]), mk(V.N_AND, leaves[2], leaves[3])
    hd = root
    for n_, nd in enumerate(nodes):
        hd |= nd << (3 + 8 * n_)
        hd |= int(rng.choice([0, 1, 2], p=[0.6, 0.3, 0.1])) << (27 + 4 * n_) | int(rng.choice([0, 1, 2], p=[0.6, 0.3, 0.1])) << (29 + 4 * n_)
    hd |= int(rng.integers(1, 1 << 16)) << 39 | carried_hdr << 55
    words[0] = hd
    ids = lambda: sum(1 << int(i) for i in rng.choice(K, int(rng.integers(0, min(K, 5) + 1)), replace=False))
    for k in range(4):
        pre = int(rng.choice([0, carried_hdr, int(rng.integers(0, K)) + 1]))
        words[V.IW_LEAF + k] = int(rng.integers(0, 4)) | int(rng.integers(0, 1 << 18)) << 2 | int(rng.integers(0, 2)) << 20 | pre << 21 | int(rng.integers(0, 2)) << 28
        words[V.IW_SET + 2 * k], words[V.IW_SET + 2 * k + 1] = ids(), ids()
    for j in range(8):
        full = rng.random() < 0.08                                           # a full list: the next object to leave raises ERR_TRACKED
        s = 0
        for q in range(4):
            v = 0xFFFF if (not full and rng.random() < 0.7) else (int(rng.integers(1, H - 1)) * W + int(rng.integers(1, W - 1)))
            s |= v << (16 * q)
        words[V.IW_STALE + j] = s
    for i in range(64):
        words[V.IW_POS + i // 4] |= pos[i] << (16 * (i % 4))
    words[V.IW_MISSION], words[V.IW_MISSION + 1] = int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 44))
    return grid, (ax, ay, d, carry_t, carry_c), act, words


@pytest.mark.parametrize("done_actions", [0, 1, 2])          # 2: AndInstr's enum-identity branch (verifier.py:561)
def test_verifier_equals_the_sequential_restatement_on_random_reachable_records(done_actions):
    L = B.load()
    rng = np.random.default_rng(7 + done_actions)
    n = 6000
    grids = np.zeros((n, W, H, 3), np.uint8)
    agents = np.zeros((n, 8), np.int32)
    acts = np.zeros(n, np.uint8)
    recs = np.zeros((n, 40), np.uint64)
    cases = []
    for i in range(n):
        g, ag, act, words = _case(rng, done_actions)
        grids[i] = g; agents[i, :5] = ag; acts[i] = act
        recs[i] = np.array(words, dtype=np.uint64)
        cases.append((g, ag, act, words))
    status = np.zeros(n, np.int32); ms = np.zeros(n, np.int32); err = np.zeros(n, np.uint32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert L.mg_selftest_verify(W, H, n, done_actions, p(grids), p(agents), p(acts), p(recs), p(status), p(ms), p(err)) == 0
    seen = {"status": set(), "err": 0, "picked": 0, "dropped": 0, "changed_leaf": 0}
    for i, (g, ag, act, words) in enumerate(cases):
        st, m, e, new = V.verify_step(words, g, W, H, ag, act, done_actions)
        got = [int(w) for w in recs[i]]
        assert (int(status[i]), int(ms[i]), int(err[i]) & V.ERR_TRACKED) == (st, m, e), (i, act, ag, hex(words[0]))
        bad = [k for k in range(40) if got[k] != new[k]]
        assert not bad, (i, act, ag, bad, [hex(got[k]) for k in bad], [hex(new[k]) for k in bad])
        seen["status"].add(st); seen["err"] += e != 0
        seen["picked"] += (new[0] >> 55) & 127 != (words[0] >> 55) & 127
        seen["changed_leaf"] += any(new[V.IW_LEAF + k] != words[V.IW_LEAF + k] for k in range(4))
    # the sample must have exercised what it is there for
    assert seen["status"] == {0, 1, 2} and seen["err"] > 5 and seen["picked"] > n // 6 and seen["changed_leaf"] > n // 4, seen
