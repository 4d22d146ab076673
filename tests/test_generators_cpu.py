"""The lane-per-episode map generators on the CPU (no GPU needed): mg_selftest_generate runs generate_one_lane (minigrid_amd/csrc/mg_genlane.h,
mg_gen.h -- the reference's _gen_grid restated on one lane's byte grid and numpy-exact PCG64 stream; the per-lane body of k_refill_lane /
k_generate_lane, which draw every spare episode of the four BASELINE.json configs) compiled for the host, into host arrays laid out like the
device's spare ring: from reset(seed)'s seeded stream, four consecutive episodes per env (= four ring slots) must reproduce the oracle's
reset()s -- which are pinned to the reference's own generated episodes (tests/golden/gen_*.npz, tests/test_oracle_golden.py) -- cell by cell,
with the agent's pose, the object it starts with in its hands, the mission (id, or the instruction record's sentence) and the stream position
after every episode (a draw too many or too few anywhere shows up there at the latest; read from the ring's stream snapshots).  Every
registered id: the product library's lane kernels serve the single-room levels, the MG_LANE_WIDE build all of them (DynamicObstacles draws
inside its step kernel: tests/test_abi_cpu.py)."""
import ctypes as C

import numpy as np
import pytest

import importlib

from minigrid_amd import _binding as B

R = importlib.import_module("minigrid_amd.registry")      # (the package re-exports the `registry` dict under the module's name)

# lane_gen_kind() (mg_genlane.h: what k_refill_lane serves on the device) + the generators that are templated on the grid type and pinned here
# ahead of being switched over on the device: GoToDoor 8, Unlock / UnlockPickup / BlockedUnlockPickup 9-11, RedBlueDoors 12, Memory 13,
# KeyCorridor 14 (and BabyAI's, 30), LockedRoom 21, Playground 22, PickupDist / OneRoom 24 25 27, OpenRedDoor 26, FindObj 28, UnlockLocal 29,
# ObstructedMaze 31, PutNear 32, the multi-room BabyAI levels 33-49, and the sentence levels 50-53 (OpenTwoDoors / OpenDoorsOrder / MoveTwoAcross /
# LevelGen = GoToSeq, Synth*, MiniBoss, BossLevel*), whose instruction record is compared through its mission sentence
SENTENCE_KINDS = {50, 51, 52, 53}
LANE_KINDS = ({0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20} | {8, 9, 10, 11, 12, 13, 14, 30} | {21, 22, 24, 25, 26, 27, 28, 29, 31, 32}
              | set(range(33, 50)) | SENTENCE_KINDS | {23})          # 23: MultiRoom
IDS = sorted(i for i, s_ in R.registry.items() if s_.env_kind in LANE_KINDS)


FLAG_SHOW_TAKEN = 16          # mg_device.h: the episode starts with an object in the agent's hands (PutNext start_carrying)


def _cfg(s, n):
    return B.MgConfig(abi_version=B.MG_ABI_VERSION, env_kind=s.env_kind, width=s.width, height=s.height, max_steps=s.max_steps,
                      see_through_walls=int(s.see_through_walls), agent_view_size=7, obs_mode=0, autoreset_mode=0, rng_mode=0, num_envs=n,
                      agent_start_x=s.agent_start[0], agent_start_y=s.agent_start[1], agent_start_dir=s.agent_start[2],
                      num_crossings=s.num_crossings, obstacle_type=s.obstacle_type, num_dists=s.num_dists, strip2_row=s.strip2_row,
                      room_size=s.room_size, random_length=int(s.random_length))


@pytest.mark.parametrize("env_id", IDS)
def test_lane_generators_on_the_host_equal_the_oracle(env_id):
    from oracle import oracle as O
    L = B.load()
    s = R.spec(env_id)
    n, E = 96, 4
    W, H = s.width, s.height
    seeds = np.arange(1000, 1000 + n, dtype=np.uint64)
    grid = np.zeros((E, n, W, H, 3), np.uint8); agent = np.zeros((E, n, 8), np.int32)
    aux = np.zeros((E, n), np.uint64); words = np.zeros((E, n, 5), np.uint64); failed = np.zeros((E, n), np.uint8)
    instr = np.zeros((E, n, 40), np.uint64)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    cfg = _cfg(s, n)
    assert L.mg_selftest_generate(C.byref(cfg), n, E, p(seeds), p(grid), p(agent), p(aux), p(words), p(failed), p(instr)) == 0
    assert not failed.any()
    orc = O.OracleVec(env_id, n)
    for ep in range(E):
        _, _, m = orc.reset(seeds=seeds if ep == 0 else None)                      # (no seed: the env's stream carries on, like an autoreset)
        g, a = orc.get_state()
        bad = np.argwhere((grid[ep] != g).reshape(n, -1).any(1)).ravel()
        assert bad.size == 0, (env_id, ep, bad[:5])
        assert (agent[ep][:, :3] == a[:, :3]).all(), (env_id, ep, "agent pose")
        assert (agent[ep][:, 3:5] == a[:, 3:5]).all(), (env_id, ep, "carried object")
        assert ((agent[ep][:, 5] & FLAG_SHOW_TAKEN != 0) == (a[:, 3] != 0)).all() and (agent[ep][:, 6] == 0).all(), (env_id, ep, "record flags / step count")
        if s.env_kind in SENTENCE_KINDS:                                            # the instruction tree, as the sentence the reference prints
            from minigrid_amd.sentence import decode
            got = [decode(int(w[37]), int(w[38])) for w in instr[ep]]
            want = list(orc.mission_strings())
            assert got == want, (env_id, ep, [(a_, b_) for a_, b_ in zip(got, want) if a_ != b_][:3])
        else:
            assert (agent[ep][:, 7] == np.asarray(m).astype(np.int64)).all(), (env_id, ep, "mission id")
        assert (words[ep] == orc.get_rng()).all(), (env_id, ep, "stream position")
    if s.env_kind in (3, 16, 17, 18, 19):                                           # GoTo levels: the tracked positions = the described objects' cells
        assert (aux[E - 1] != 0).all()
