#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: runs product-level parity cases against a library built for the host SIMT emulator (tests/emu/build_emu.py) in THIS process
(the library is chosen once per process: MINIGRID_AMD_LIB) and prints one JSON line per case.  tests/test_emu_cpu.py drives it.

    MINIGRID_AMD_LIB=/tmp/minigrid_emu_build/libminigrid_emu.so MINIGRID_AMD_NO_TORCH=1 python tests/emu/run_cases.py '<json list of cases>'

A case: {"env": id, "n": envs, "launches": [T, ...], "full": bool, "max_steps": k | null, "knobs": {"MG_...": "v"}, "stepped": steps,
"autoreset": "next_step" | "same_step", "obs_mode": ..., "view": ViewSizeWrapper's size, "spare_ring": R}: reset(seed=0), then fused launches of the given lengths under the device's Philox policy -- every slot's
image, reward bytes, flags, direction, mission (id or sentence) against the oracle -- then `stepped` single steps with caller actions, then the
final state and every env's stream position."""
import json
import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def run_case(c):
    import minigrid_amd as mg
    from par_oracle import ParOracle
    from oracle import oracle as O
    for k, v in c.get("knobs", {}).items():
        os.environ[k] = str(v)
    try:
        env_id, n, full = c["env"], int(c["n"]), bool(c.get("full", False))
        kw = {} if c.get("max_steps") is None else {"max_steps": int(c["max_steps"])}
        mode = c.get("autoreset", "next_step")
        obs_mode = c.get("obs_mode", "full" if full else "partial")               # partial | full | onehot | symbolic | rgb | rgb_partial
        view = int(c.get("view", 7))
        mk = dict(kw)
        if c.get("spare_ring"):
            mk["spare_ring"] = int(c["spare_ring"])
        env = mg.make_vec(env_id, n, obs_mode=obs_mode, autoreset_mode=mode, agent_view_size=view, **mk)
        orc = ParOracle(env_id, n, full, threads=1, obs=obs_mode, view_size=view, **kw)
        ar = 2 if mode == "same_step" else 1
        obs, _ = env.reset(seed=0)
        assert (obs["image"] == orc.reset(0)[0]).all(), "reset image"
        if env.sentence:
            assert (np.asarray(obs["mission"]) == orc.vecs[0].mission_strings()).all(), "reset mission sentence"
        seed, t, fin = 2, 0, 0
        for T in c.get("launches", []):
            env.rollout(T, action_seed=seed, fused=True)
            for k in reversed(range(T)):
                act = O.philox_actions(seed, t, n)
                oo, orew, oterm, otrunc, od, om = orc.vecs[0].step(act, autoreset=ar); t += 1
                img, rew, term, trunc, d, m, a = env.trajectory(k)
                what = (env_id, "T", T, "slot", k)
                assert (a == act).all(), (what, "recorded actions")
                bad = np.argwhere((img != oo).reshape(n, -1).any(1)).ravel()
                assert bad.size == 0, (what, "image", bad[:5].tolist())
                assert rew.tobytes() == orew.tobytes(), (what, "reward")
                assert (term == oterm).all() and (trunc == otrunc).all(), (what, "flags")
                assert (d == od).all(), (what, "direction")
                if env.sentence:
                    if k == 0:
                        assert (np.asarray(env.trajectory_missions(0)) == orc.vecs[0].mission_strings()).all(), (what, "mission sentence")
                else:
                    assert (m == om).all(), (what, "mission id")
                fin += int(term.sum()) + int(trunc.sum())
        rng = np.random.default_rng(5)
        for s in range(int(c.get("stepped", 0))):
            act = rng.choice(7, size=n, p=[0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05]).astype(np.uint8)
            obs, rew, term, trunc, _ = env.step(act)
            oo, orew, oterm, otrunc, od, om = orc.vecs[0].step(act, autoreset=ar)
            assert (obs["image"] == oo).all(), (env_id, "stepped", s, "image")
            assert rew.tobytes() == orew.tobytes() and (term == oterm).all() and (trunc == otrunc).all(), (env_id, "stepped", s, "scalars")
            fin += int(term.sum()) + int(trunc.sum())
        g1, a1 = env.get_state(); g2, a2 = orc.get_state()
        assert (g1 == g2).all() and (a1[:, :7] == a2[:, :7]).all(), "final state"
        assert (env.get_rng_state() == orc.get_rng()).all(), "stream positions"
        env.close() if hasattr(env, "close") else None
        return {"ok": True, "episodes": fin}
    finally:
        for k in c.get("knobs", {}):
            os.environ.pop(k, None)


if __name__ == "__main__":
    from minigrid_amd import _binding as B
    info = B.load().mg_build_info().decode()
    assert "emulator=1" in info, f"not an emulator build: {info}"
    for c in json.loads(sys.argv[1]):
        try:
            r = run_case(c)
        except Exception as ex:      # noqa: BLE001  (reported per case)
            r = {"ok": False, "error": repr(ex)[:600], "where": traceback.format_exc().strip().splitlines()[-3:]}
        print(json.dumps({"case": c, **r}), flush=True)
