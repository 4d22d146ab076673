"""MG_RNG_PHILOX (the device-side Philox-keyed map generator BASELINE.json's north star names): SURVEY.md §8(c)(2),(3).
Philox layouts differ from numpy's PCG64 layouts by construction, so parity is established the way §8(c) prescribes:
 (3) the generator's marginals against 10^4 oracle resets (chi-square homogeneity tests), and
 (2) state injection -- the device-generated episodes are handed to the oracle (get_state -> set_state) and both are
     stepped with identical actions for 300 steps, every output compared bit for bit."""
import numpy as np
import pytest

def _on_emu():
    import os
    return os.environ.get("MINIGRID_AMD_EMU_RERUN") == "1"


pytestmark = pytest.mark.gpu

N_CHI = 10240


def _chi2_same(a, b, what, p=1e-6):
    """two-sample chi-square homogeneity test of two count vectors over the same categories"""
    from scipy.stats import chi2
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    keep = (a + b) > 0
    a, b = a[keep], b[keep]
    if keep.sum() < 2:                      # a constant feature (e.g. the number of lava cells of one river): both must have it
        assert a.sum() > 0 and b.sum() > 0, what
        return
    ka, kb = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    stat = (((ka * a - kb * b) ** 2) / (a + b)).sum()
    lim = chi2.isf(p, len(a) - 1)
    assert stat < lim, (what, stat, lim, a[:12], b[:12])


def _device_states(env_id, n, rounds=1, **kw):
    import minigrid_amd as mg
    env = mg.make_vec(env_id, n, rng="philox", **kw)
    grids, agents = [], []
    env.reset(seed=12345)
    for r in range(rounds):
        g, a = env.get_state()
        grids.append(g); agents.append(a)
        if r + 1 < rounds:
            env.reset()                      # the next episode of every env's stream (consumes the spare ring)
    env.close()
    return np.concatenate(grids), np.concatenate(agents)


def _oracle_states(env_id, n, seed0=777):
    from oracle import oracle as O
    orc = O.OracleVec(env_id, n)
    orc.reset(seeds=np.arange(seed0, seed0 + n, dtype=np.uint64))
    return orc.get_state()


def _pos_of(grid, typ, color=None):
    """cell index x*H+y of the FIRST cell of this type (and colour) per env, -1 if none"""
    n = grid.shape[0]
    m = grid[..., 0] == typ
    if color is not None:
        m &= grid[..., 1] == color
    flat = m.reshape(n, -1)
    return np.where(flat.any(1), flat.argmax(1), -1)


def _hist(v, k):
    return np.bincount(np.asarray(v) + 1, minlength=k + 1)       # slot 0 = "none"


def test_philox_doorkey_marginals_match_the_reference_generator():
    gd, ad = _device_states("MiniGrid-DoorKey-8x8-v0", N_CHI)
    go, ao = _oracle_states("MiniGrid-DoorKey-8x8-v0", N_CHI)
    for name, typ in (("door", 4), ("key", 5)):
        _chi2_same(_hist(_pos_of(gd, typ), 64), _hist(_pos_of(go, typ), 64), f"doorkey {name} position")
    _chi2_same(_hist(ad[:, 0] * 8 + ad[:, 1], 64), _hist(ao[:, 0] * 8 + ao[:, 1], 64), "doorkey agent position")
    _chi2_same(_hist(ad[:, 2], 4), _hist(ao[:, 2], 4), "doorkey agent dir")
    # joint of (split column, door row): the door's x is the wall column
    _chi2_same(_hist(_pos_of(gd, 4) // 8, 8), _hist(_pos_of(go, 4) // 8, 8), "doorkey split column")
    # invariants of every generated map (doorkey.py:74-99)
    door = _pos_of(gd, 4); key = _pos_of(gd, 5)
    assert (door >= 0).all() and (key >= 0).all()
    assert (key // 8 < door // 8).all() and (ad[:, 0] < door // 8).all()
    assert (gd[np.arange(len(gd)), door // 8, door % 8, 2] == 2).all() and (gd[np.arange(len(gd)), door // 8, door % 8, 1] == 4).all()


@pytest.mark.parametrize("env_id", ["MiniGrid-LavaCrossingS9N1-v0", "MiniGrid-SimpleCrossingS9N2-v0"])
def test_philox_crossing_marginals_match_the_reference_generator(env_id):
    gd, _ = _device_states(env_id, N_CHI)
    go, _ = _oracle_states(env_id, N_CHI)
    obst = 9 if "Lava" in env_id else 2

    def river_signature(g):
        m = (g[:, 1:-1, 1:-1, 0] == obst).reshape(len(g), -1)          # interior obstacle pattern
        # a compact categorical: (#obstacle cells, index of first, index of last)
        first = np.where(m.any(1), m.argmax(1), 0)
        last = np.where(m.any(1), m.shape[1] - 1 - m[:, ::-1].argmax(1), 0)
        return first, last, m.sum(1), m
    fd, ld, cd, md = river_signature(gd)
    fo, lo, co, mo = river_signature(go)
    _chi2_same(_hist(fd, 49), _hist(fo, 49), f"{env_id} first obstacle cell")
    _chi2_same(_hist(ld, 49), _hist(lo, 49), f"{env_id} last obstacle cell")
    _chi2_same(_hist(cd, 49), _hist(co, 49), f"{env_id} obstacle count")
    # per-cell frequencies: the cells of one river are perfectly correlated, so compare them as rates, not as a chi-square
    assert np.abs(md.mean(0) - mo.mean(0)).max() < 0.03, np.abs(md.mean(0) - mo.mean(0)).max()


def test_philox_gotoredball_marginals_match_the_reference_generator():
    gd, ad = _device_states("BabyAI-GoToRedBall-v0", N_CHI)
    go, ao = _oracle_states("BabyAI-GoToRedBall-v0", N_CHI)
    _chi2_same(_hist(ad[:, 0] * 8 + ad[:, 1], 64), _hist(ao[:, 0] * 8 + ao[:, 1], 64), "gotoredball agent position")
    _chi2_same(_hist(ad[:, 2], 4), _hist(ao[:, 2], 4), "gotoredball agent dir")
    _chi2_same(_hist(ad[:, 7], 2), _hist(ao[:, 7], 2), "gotoredball mission (the / a red ball)")

    def objects(g):
        inner = g[:, 1:-1, 1:-1]
        t, c = inner[..., 0].reshape(len(g), -1), inner[..., 1].reshape(len(g), -1)
        isobj = (t >= 5) & (t <= 7)
        tc = np.where(isobj, (t - 5) * 6 + c, -1)
        return np.bincount(tc[isobj].ravel(), minlength=18), isobj.sum(0), isobj.sum(1)
    tcd, pd_, nd = objects(gd)
    tco, po, no = objects(go)
    _chi2_same(tcd, tco, "gotoredball object (type, colour) counts")
    _chi2_same(pd_, po, "gotoredball object positions")
    _chi2_same(_hist(nd, 9), _hist(no, 9), "gotoredball objects per map")
    assert (nd == 8).all()                                            # the ball + 7 distractors, none lost


def test_philox_empty_random_and_later_episodes_of_the_stream():
    # Empty-Random: place_agent(); three consecutive episodes per env (the spare ring in Philox mode)
    gd, ad = _device_states("MiniGrid-Empty-Random-6x6-v0", 4096, rounds=3)
    go, ao = _oracle_states("MiniGrid-Empty-Random-6x6-v0", 3 * 4096)
    _chi2_same(_hist(ad[:, 0] * 6 + ad[:, 1], 36), _hist(ao[:, 0] * 6 + ao[:, 1], 36), "empty-random agent position")
    _chi2_same(_hist(ad[:, 2], 4), _hist(ao[:, 2], 4), "empty-random agent dir")
    a0, a1 = ad[:4096, :3], ad[4096:8192, :3]
    assert (a0 != a1).any(1).mean() > 0.9                              # episodes of one stream differ


@pytest.mark.parametrize("env_id,full", [("MiniGrid-Empty-Random-6x6-v0", False), ("MiniGrid-DoorKey-8x8-v0", False),
                                         ("MiniGrid-LavaCrossingS9N1-v0", True), ("BabyAI-GoToRedBall-v0", False)])
def test_philox_generated_episodes_step_like_the_reference_after_state_injection(env_id, full):
    """device-generated (Philox) episodes injected into the oracle; 300 identical steps, every output."""
    import minigrid_amd as mg
    from oracle import oracle as O
    n = 4096 if not _on_emu() else 200       # (tests/test_emu_gpu_suite_cpu.py re-runs this on the host emulator: the same kernels, fewer envs)
    mode = "full" if full else "partial"
    env = mg.make_vec(env_id, n, rng="philox", obs_mode=mode, autoreset_mode="disabled")
    orc = O.OracleVec(env_id, n, full_obs=full)
    obs, _ = env.reset(seed=99)
    orc.reset(seeds=np.arange(n, dtype=np.uint64))                       # any state: overwritten by the injection
    orc.set_state(*env.get_state())
    rng = np.random.default_rng(5)
    probs = [0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05]
    nterm = 0
    for t in range(300):
        a = rng.choice(7, size=n, p=probs).astype(np.uint8)
        obs, rew, term, trunc, _ = env.step(a)
        oo, orew, oterm, otrunc, od, om = orc.step(a, autoreset=0)
        assert (obs["image"] == oo).all(), (env_id, t)
        assert rew.tobytes() == orew.tobytes() and (term == oterm).all() and (trunc == otrunc).all(), (env_id, t)
        assert (obs["direction"] == od).all()
        nterm += int(term.sum())
    if ("Empty" in env_id or "Lava" in env_id) and not _on_emu():
        assert nterm > n // 4
    g1, a1 = env.get_state(); g2, a2 = orc.get_state()
    assert (g1 == g2).all() and (a1[:, :6] == a2[:, :6]).all()
    env.close()


def test_philox_fused_rollout_is_deterministic_and_matches_stepping():
    """Philox mode through the fused path: two handles agree, and fused == step-by-step with the recorded actions."""
    import minigrid_amd as mg
    n = 3000 if not _on_emu() else 200
    a = mg.make_vec("BabyAI-GoToRedBall-v0", n, rng="philox", traj_slots=16)
    b = mg.make_vec("BabyAI-GoToRedBall-v0", n, rng="philox")
    a.reset(seed=8); b.reset(seed=8)
    for c in range(8):
        a.rollout(16, action_seed=4, fused=True)
        for k in reversed(range(16)):
            img, rew, term, trunc, d, m, act = a.trajectory(k)
            obs, r2, t2, u2, _ = b.step(act)
            assert (img == obs["image"]).all() and rew.tobytes() == r2.tobytes() and (term == t2).all() and (trunc == u2).all()
    ga, aa = a.get_state(); gb, ab = b.get_state()
    assert (ga == gb).all() and (aa == ab).all()
    a.close(); b.close()
