"""CPU-side checks of the product library: it builds, loads, exports the whole C ABI, its host-callable inline
helpers are exact, and the product never falls back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from minigrid_amd import _binding as B


def test_library_builds_and_exports_every_declared_symbol():
    L = B.load()
    header = open(os.path.join(ROOT, "include", "minigrid_hip.h")).read()
    declared = set(re.findall(r"MG_API\s+(?:const char\*|int)\s+(mg_\w+)\s*\(", header))
    assert declared == set(B.SYMBOLS), declared ^ set(B.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mg_abi_version() == B.MG_ABI_VERSION


def test_config_struct_layout_matches_header():
    # 20 int32, double at 80, 2 int32, int64 at 96, 5 int32 (+ 4 bytes of tail padding) = 128 bytes
    assert C.sizeof(B.MgConfig) == 128 and B.MgConfig.babyai_done_actions.offset == 120
    assert B.MgConfig.death_cost.offset == 80
    assert B.MgConfig.env_index_base.offset == 96
    assert B.MgConfig.tile_size.offset == 104 and B.MgConfig.rgb_highlight.offset == 108
    assert B.MgConfig.spare_ring.offset == 112 and B.MgConfig.traj_slots.offset == 116
    assert C.sizeof(B.MgOutputs) == 120 and B.MgOutputs.scalar_stride.offset == 112 and B.MgOutputs.action.offset == 64 and B.MgOutputs.max_fused_steps.offset == 96 and B.MgOutputs.sentence.offset == 104


@pytest.mark.parametrize("obe", [147, 243, 27, 75, 363, 507, 675, 192, 980, 49, 64, 361 * 3, 5, 6, 7, 8, 9])
def test_observation_stream_packer_reproduces_the_byte_stream(obe):
    """k_step writes a wave's observations as whole aligned dwords of ONE contiguous byte stream although an env's
    obe bytes (147, 243, ...) start at any byte phase: StreamEmit (mg_kernels.h) run lane by lane on the host."""
    L = B.load()
    rng = np.random.default_rng(obe)
    for lpe in (1, 4):
        if lpe == 4 and (obe % 3 or obe // 12 < 4):
            continue                         # 4 lanes per env: three-byte cells, at least one 12-byte unit per lane
        for nenv in (1, 2, 3, 4, 5, 15, 16) if lpe == 4 else (1, 2, 3, 4, 5, 63, 64):
            src = rng.integers(0, 256, (nenv, obe), dtype=np.uint8)
            out = np.zeros(nenv * obe, np.uint8)
            assert L.mg_selftest_stream(obe, nenv, lpe, src.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(out, src.reshape(-1)), (obe, nenv, lpe)


def _vis_row_literal(m, t):
    """core/grid.py:296-321 for one row j > 0: returns (row mask after both sweeps, bits set in row j-1)."""
    mask = [(m >> i) & 1 for i in range(7)]
    up = [0] * 7
    for i in range(0, 6):
        if not mask[i]:
            continue
        if not (t >> i) & 1:
            continue
        mask[i + 1] = 1
        up[i + 1] = 1
        up[i] = 1
    for i in reversed(range(1, 7)):
        if not mask[i]:
            continue
        if not (t >> i) & 1:
            continue
        mask[i - 1] = 1
        up[i - 1] = 1
        up[i] = 1
    return sum(b << i for i, b in enumerate(mask)), sum(b << i for i, b in enumerate(up))


def test_vis_row_bit_parallel_equals_reference_loops_exhaustively():
    L = B.load()
    mo, uo = C.c_uint32(), C.c_uint32()
    for m in range(128):
        for t in range(128):
            assert L.mg_selftest_vis_row(m, t, C.byref(mo), C.byref(uo)) == 0
            assert (mo.value, uo.value) == _vis_row_literal(m, t), (m, t)


@pytest.mark.parametrize("T", [64, 100, 256, 324, 484, 640, 2560, 65535])
def test_reward_lut_is_bit_exact_python_float_arithmetic(T):
    L = B.load()
    out = np.zeros(T + 1, np.float64)
    assert L.mg_selftest_reward_lut(T, out.ctypes.data_as(C.c_void_p)) == 0
    want = np.array([1 - 0.9 * (t / T) for t in range(T + 1)], np.float64)
    assert out.tobytes() == want.tobytes()


def test_cell_code_roundtrip_matches_worldobj_decode_encode():
    """cell_from_triple/cell_triple == WorldObj.decode(...).encode() (core/world_object.py:65-102,196-212)."""
    L = B.load()
    code, tri = C.c_uint32(), C.c_uint32()
    for t in range(11):
        for c in range(6):
            for s in range(3):
                L.mg_selftest_pack_cell(t, c, s, C.byref(code), C.byref(tri))
                got = (tri.value & 255, (tri.value >> 8) & 255, tri.value >> 16)
                if t in (0, 1, 10):
                    want = (1, 0, 0)              # decode -> None -> Grid.encode writes (1,0,0)
                elif t == 8:
                    want = (8, 1, 0)              # Goal() is green
                elif t == 9:
                    want = (9, 0, 0)              # Lava() is red
                elif t == 4:
                    want = (4, c, s)
                else:
                    want = (t, c, 0)
                assert got == want, (t, c, s, got)
                assert code.value < 256


def test_no_cpu_fallback_without_gpu():
    import minigrid_amd as mg
    if B.load().mg_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(B.MiniGridHipError):
        mg.make_vec("MiniGrid-Empty-8x8-v0", 8)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "minigrid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_registry_rows_match_oracle_table():
    import minigrid_amd as mg
    from oracle import oracle as O
    for env_id, row in mg.registry.items():
        o = O.spec(env_id)
        if o["kind"] == O.K_LEVELGEN:      # per-episode max_steps: the oracle's row holds the 1-nav value, the registry the reference env's
            assert row.max_steps % o["max_steps"] == 0, env_id
        else:
            assert row.max_steps == o["max_steps"], env_id
        assert (row.env_kind, row.width, row.height, int(row.see_through_walls)) == (o["kind"], o["width"], o["height"], o["see_through"]), env_id
        assert row.strip2_row == o.get("strip2_row", 0) or o["kind"] not in (O.K_LEVELGEN, O.K_OPENTWODOORS, O.K_OPENDOORSORDER), env_id
        assert list(row.missions) == o["missions"]
        assert row.num_crossings == o.get("num_crossings", 0) and row.num_dists == o.get("num_dists", 0)


def _vis_row_literal_n(m, t, V):
    mask = [(m >> i) & 1 for i in range(V)]
    up = [0] * V
    for i in range(0, V - 1):
        if mask[i] and (t >> i) & 1:
            mask[i + 1] = 1; up[i + 1] = 1; up[i] = 1
    for i in reversed(range(1, V)):
        if mask[i] and (t >> i) & 1:
            mask[i - 1] = 1; up[i - 1] = 1; up[i] = 1
    return sum(b << i for i, b in enumerate(mask)), sum(b << i for i, b in enumerate(up))


@pytest.mark.parametrize("V", [3, 5, 7, 9, 11, 13, 15])
def test_vis_row_n_equals_reference_loops(V):
    """ViewSizeWrapper widths: exhaustive up to V = 7, 20 000 random (mask, transparency) pairs above."""
    L = B.load()
    mo, uo = C.c_uint32(), C.c_uint32()
    if V <= 7:
        pairs = [(m, t) for m in range(1 << V) for t in range(1 << V)]
    else:
        rng = np.random.default_rng(V)
        pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, 1 << V, 20000), rng.integers(0, 1 << V, 20000))]
        pairs += [(1 << (V // 2), (1 << V) - 1), ((1 << V) - 1, 0), (1, (1 << V) - 1), (1 << (V - 1), (1 << V) - 1)]
    for m, t in pairs:
        assert L.mg_selftest_vis_row_n(V, m, t, C.byref(mo), C.byref(uo)) == 0
        assert (mo.value, uo.value) == _vis_row_literal_n(m, t, V), (V, m, t)


def test_dict_observation_space_vocabulary_doctest():
    # reference doctest minigrid/wrappers.py:442-447 (LavaCrossingS11N5 mission)
    from minigrid_amd.mission_vocab import MAX_WORDS_IN_MISSION, minigrid_words, string_to_indices
    idx = string_to_indices("avoid the lava and get to the green goal square")
    assert idx[:10] == [19, 31, 17, 36, 20, 38, 31, 2, 15, 35] and len(idx) == MAX_WORDS_IN_MISSION and idx[10:] == [0] * 40
    assert len(minigrid_words()) == 51
    with pytest.raises(ValueError):
        string_to_indices("avoid the dragon")


def _cfg(**kw):
    base = dict(abi_version=B.MG_ABI_VERSION, env_kind=0, width=8, height=8, max_steps=256, see_through_walls=1,
                agent_view_size=7, obs_mode=0, autoreset_mode=0, rng_mode=0, num_envs=64, agent_start_x=1,
                agent_start_y=1, agent_start_dir=0)
    base.update(kw)
    return B.MgConfig(**base)


@pytest.mark.parametrize("bad", [dict(num_envs=0), dict(width=2), dict(width=26), dict(agent_view_size=4),
                                 dict(agent_view_size=17), dict(max_steps=0), dict(env_kind=99), dict(obs_mode=7),
                                 dict(abi_version=0), dict(no_death_mask=1 << 8),
                                 dict(env_kind=2, width=8, height=8),                    # Crossing needs an odd size
                                 dict(env_kind=13, width=8, height=8),                   # Memory needs an odd size
                                 dict(env_kind=12, width=8, height=8),                   # RedBlueDoors: width = 2 * height
                                 dict(env_kind=9, width=11, height=6, room_size=5),      # Unlock: inconsistent room size
                                 dict(env_kind=19, width=9, height=9, num_dists=3),      # GoToLocal: room_size <= 8
                                 dict(obs_mode=5, tile_size=0), dict(obs_mode=4, tile_size=65),     # RGB: tile_size in 1..64
                                 dict(autoreset_mode=3), dict(autoreset_mode=2, env_kind=15, obs_mode=1),      # SAME_STEP of DynamicObstacles: the 7x7 view only
                                 dict(env_kind=23, width=25, height=25, room_size=10, num_crossings=2, num_dists=7),   # MultiRoom: <= 6 rooms
                                 dict(env_kind=23, width=25, height=25, room_size=3, num_crossings=2, num_dists=2),    # maxRoomSize >= 4
                                 dict(env_kind=28, width=13, height=13, room_size=6),    # FindObj: 3*(room_size-1)+1
                                 dict(env_kind=26, width=9, height=6, room_size=5),      # OpenRedDoor: height = room_size
                                 dict(env_kind=24, width=7, height=8),                   # PickupDist: one square room
                                 dict(env_kind=29, width=22, height=22, room_size=8, num_dists=9),      # UnlockLocal: <= 8 distractors
                                 dict(env_kind=20, width=8, height=8, num_dists=0),      # GoToObject: numObjs >= 1
                                 dict(env_kind=33, width=22, height=22, room_size=8, num_dists=22),     # BabyAI GoTo: <= 21 distractors
                                 dict(env_kind=36, width=16, height=6, room_size=6),     # UnlockPickup: 1 x 2 rooms
                                 dict(env_kind=39, width=11, height=6, room_size=6),     # KeyInBox: 3 x 3 rooms
                                 dict(env_kind=46, width=8, height=8, room_size=8, num_dists=1),        # PutNextLocal: >= 2 objects
                                 dict(env_kind=47, width=13, height=7, room_size=7, num_dists=5),       # PutNext: <= 4 objects per room
                                 dict(env_kind=49, width=22, height=22, room_size=8, num_crossings=3),  # OpenDoor: select_by 0..2
                                 dict(env_kind=51, width=16, height=16, room_size=6, num_dists=5),      # OpenDoorsOrder: 2..4 doors
                                 dict(env_kind=52, width=15, height=15, room_size=8, num_dists=9),      # MoveTwoAcross: 1 x 2 rooms
                                 dict(env_kind=53, width=22, height=22, room_size=8, num_dists=18, num_crossings=0),   # LevelGen: some action kind
                                 dict(env_kind=53, width=22, height=22, room_size=8, num_dists=18, num_crossings=0x1F, strip2_row=101),  # probability in percent
                                 dict(env_kind=31, width=16, height=16, room_size=5, num_dists=1),      # ObstructedMaze: room_size 6
                                 dict(env_kind=32, width=9, height=9, num_dists=2)])     # PutNear: size 5..8
def test_mg_create_rejects_bad_configs_before_touching_a_device(bad):
    """Validation comes first (MG_ERR_INVALID with a message); only a valid config gets as far as the device check,
    which on this GPU-less box answers MG_ERR_NO_DEVICE -- there is no CPU fallback to fall into."""
    L = B.load()
    h = C.c_void_p()
    cfg = _cfg(**bad)
    assert L.mg_create(C.byref(cfg), -1, None, C.byref(h)) == B.MG_ERR_INVALID
    assert not h.value and len(L.mg_last_error(None)) > 10


def test_mg_create_valid_config_needs_a_device():
    L = B.load()
    if L.mg_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    cfg = _cfg()
    assert L.mg_create(C.byref(cfg), -1, None, C.byref(h)) == B.MG_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.mg_last_error(None)
    import minigrid_amd as mg
    with pytest.raises(B.MiniGridHipError):
        mg.make_vec("MiniGrid-Empty-8x8-v0", 64)


def test_registry_rows_are_consistent():
    import minigrid_amd as mg
    assert len(mg.registry) == 172
    for env_id, s in mg.registry.items():
        assert s.id == env_id and 3 <= s.width <= 25 and 3 <= s.height <= 25 and 1 <= s.max_steps <= 65535 and len(s.missions) >= 1
        assert s.entry_point.startswith("minigrid.envs")


def test_registry_and_oracle_tables_match_the_reference_registry():
    """tests/golden/reference_registry.json is written by oracle/make_golden.py from the reference's own registry and
    instantiated envs: entry point, kwargs, grid size, max_steps, see_through_walls of every id."""
    import json
    import os

    import minigrid_amd as mg
    from oracle import oracle as O
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_registry.json")))
    assert len(ref) == 172
    for env_id, s in mg.registry.items():
        r = ref[env_id]
        assert s.entry_point == r["entry_point"], env_id
        assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in s.kwargs.items()} == r["kwargs"], (env_id, s.kwargs, r["kwargs"])
        assert (s.width, s.height, s.max_steps, bool(s.see_through_walls)) == (r["width"], r["height"], r["max_steps"], r["see_through_walls"]), env_id
        assert r["agent_view_size"] == 7
        for mid, text in r["missions_seen"].items():      # the mission strings the reference produced, by mission id
            assert s.missions[int(mid)] == text, (env_id, mid)
    for env_id, r in ref.items():
        o = O.spec(env_id)
        for mid, text in r["missions_seen"].items():
            assert o["missions"][int(mid)] == text, (env_id, mid)
        assert (o["width"], o["height"], bool(o["see_through"])) == (r["width"], r["height"], r["see_through_walls"]), env_id
        if o["kind"] != O.K_LEVELGEN:                     # LevelGen levels: max_steps depends on the drawn instruction
            assert o["max_steps"] == r["max_steps"], (env_id, o["max_steps"], r["max_steps"])


def test_library_tile_atlas_matches_every_reference_tile():
    """mg_render_tiles is the host routine mg_create fills the RGB atlas with (mg_tiles.h): all 510 tiles x 4 tile sizes
    against tiles rendered by the reference's Grid.render_tile (tests/golden/rgb_atlas.npz)."""
    from conftest import golden
    L = B.load()
    g = golden("rgb_atlas.npz")
    for ts in g["tile_sizes"]:
        want = g[f"tiles{ts}"]
        out = np.zeros(want.shape, np.uint8)
        assert L.mg_render_tiles(int(ts), out.ctypes.data) == 0
        assert (out == want).all(), ts
    assert L.mg_render_tiles(0, out.ctypes.data) != 0


def test_golden_generator_id_lists_match_the_test_lists():
    """oracle/make_golden.py (which needs the reference to import) and tests/conftest.py each spell the env id lists out:
    read the generator's lists with ast and compare."""
    import ast
    import os

    import conftest
    src = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "make_golden.py")).read()
    lists = {}
    for node in ast.parse(src).body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            if name in ("MAIN_IDS", "EXTRA_IDS", "WIDE_IDS", "WIDE2_IDS", "SENTENCE_IDS", "ORACLE_ONLY_IDS"):
                lists[name] = ast.literal_eval(node.value)
    for name, ids in lists.items():
        assert ids == getattr(conftest, name), name
    assert set(lists) == {"MAIN_IDS", "EXTRA_IDS", "WIDE_IDS", "WIDE2_IDS", "SENTENCE_IDS", "ORACLE_ONLY_IDS"}


# ---- k_roll7's observation pipeline (minigrid_amd/csrc/mg_roll.h) on the host ----------------------------------------------

def test_vis_row_carry_equals_vis_row_exhaustively():
    """process_vis rows by carry propagation == the Kogge-Stone form (itself checked against the literal loops above)."""
    import ctypes as C
    L = B.load()
    m1, u1, m2, u2 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    for m in range(128):
        for t in range(128):
            L.mg_selftest_vis_row(m, t, C.byref(m1), C.byref(u1))
            L.mg_selftest_vis_row_carry(m, t, C.byref(m2), C.byref(u2))
            assert (m1.value, u1.value) == (m2.value, u2.value), (m, t)


def test_valu_primitives_host_forms():
    """perm_b32 / udot4 / brev32 / expand4 as the host evaluates them, against plain Python (the GPU suite compares the
    device instructions with these host forms)."""
    L = B.load()
    rng = np.random.default_rng(0)
    n = 4096
    a = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    sel = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 0x0C, 0x0D], size=(n, 4)).astype(np.uint32)
    c = (sel[:, 0] | (sel[:, 1] << 8) | (sel[:, 2] << 16) | (sel[:, 3] << 24)).astype(np.uint32)
    out = np.zeros((6, n), np.uint32)
    p = lambda x: x.ctypes.data_as(__import__("ctypes").c_void_p)
    assert L.mg_selftest_prims(n, p(a), p(b), p(c), p(out), 0) == 0
    for i in range(0, n, 37):
        v = (int(a[i]) << 32) | int(b[i])
        want = 0
        for k in range(4):
            s = int(sel[i, k])
            byte = (v >> (8 * s)) & 0xFF if s < 8 else (0 if s == 0x0C else 0xFF)
            want |= byte << (8 * k)
        assert int(out[0, i]) == want
        dot = sum(((int(a[i]) >> (8 * k)) & 0xFF) * ((int(b[i]) >> (8 * k)) & 0xFF) for k in range(4)) + int(c[i])
        assert int(out[1, i]) == dot & 0xFFFFFFFF
        assert int(out[2, i]) == int(format(int(a[i]), "032b")[::-1], 2)
        assert int(out[3, i]) == sum(0xFF << (8 * k) for k in range(4) if (int(a[i]) >> k) & 1)


@pytest.mark.parametrize("env_id,n,T", [("MiniGrid-DoorKey-8x8-v0", 200, 80), ("MiniGrid-Empty-5x5-v0", 64, 40),
                                        ("MiniGrid-LavaCrossingS9N1-v0", 130, 60), ("BabyAI-GoToRedBall-v0", 100, 60),
                                        ("MiniGrid-FourRooms-v0", 70, 80), ("MiniGrid-MultiRoom-N6-v0", 40, 60),
                                        ("MiniGrid-DistShift1-v0", 64, 40), ("MiniGrid-ObstructedMaze-1Dlhb-v0", 65, 80),
                                        ("MiniGrid-KeyCorridorS3R3-v0", 64, 80), ("MiniGrid-Unlock-v0", 64, 80)])
def test_roll7_observation_pipeline_on_the_host_equals_the_oracle(env_id, n, T):
    """mg_selftest_obs7 = the device code of k_roll7's observation (line gather, byte transposes, carry visibility rows, output-space
    encode) compiled for the host: every observation of random rollouts (doors opened, objects carried, every pose at the borders)."""
    import ctypes as C
    from oracle import oracle as O
    L = B.load()
    orc = O.OracleVec(env_id, n)
    obs, _, _ = orc.reset(seeds=np.arange(n, dtype=np.uint64))
    rng = np.random.default_rng(1)
    out = np.zeros((n, 7, 7, 3), np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    see = int(O.spec(env_id)["see_through"])
    for t in range(T):
        grid, agent = orc.get_state()
        assert L.mg_selftest_obs7(orc.W, orc.H, n, p(np.ascontiguousarray(grid)), p(np.ascontiguousarray(agent)), see, p(out)) == 0
        bad = np.argwhere((out != obs).reshape(n, -1).any(1)).ravel()
        assert bad.size == 0, (env_id, t, bad[:5], agent[bad[:1]], out[bad[0]].reshape(49, 3)[:, 0].reshape(7, 7), obs[bad[0]].reshape(49, 3)[:, 0].reshape(7, 7))
        a = rng.choice(7, size=n, p=[0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05]).astype(np.uint8)
        obs = orc.step(a)[0]


DYNOBS_IDS = ["MiniGrid-Dynamic-Obstacles-5x5-v0", "MiniGrid-Dynamic-Obstacles-Random-5x5-v0", "MiniGrid-Dynamic-Obstacles-6x6-v0",
              "MiniGrid-Dynamic-Obstacles-Random-6x6-v0", "MiniGrid-Dynamic-Obstacles-8x8-v0", "MiniGrid-Dynamic-Obstacles-16x16-v0"]


@pytest.mark.parametrize("redo", [0, 2])
@pytest.mark.parametrize("env_id", DYNOBS_IDS)
def test_dynobs_in_loop_draws_on_the_host_equal_the_oracle(env_id, redo):
    """mg_selftest_dynobs = dynobs_place (mg_dynobs.h), the per-lane loop k_roll7<GG_DYNOBS> runs for DynamicObstacles' obstacle moves and
    in-place resets, compiled for the host: from numpy's freshly seeded PCG64 words it must reproduce the oracle's grids, agent poses and
    stream positions reset after reset and step after step (the obstacle LIST order is carried by the function itself: a wrong order, a draw
    too many or a wrong rejection shows up as a diverging grid or stream).  redo = 2: every try through the draw code's rare-case path (a
    Lemire rejection candidate: once in ~10^9 tries by chance), which restores the stream and redraws with the general bounded-integer code."""
    import ctypes as C
    from oracle import oracle as O
    L = B.load()
    s = O.spec(env_id)
    n, T = 48, 160
    W, H, nob = s["width"], s["height"], s["num_dists"]
    orc = O.OracleVec(env_id, n)
    seeds = np.arange(100, 100 + n, dtype=np.uint64)
    words = np.zeros((n, 5), np.uint64)
    for i, sd in enumerate(seeds):
        st = np.random.PCG64(np.random.SeedSequence(int(sd))).state["state"]
        words[i] = [st["state"] >> 64, st["state"] & (2 ** 64 - 1), st["inc"] >> 64, st["inc"] & (2 ** 64 - 1), 0]
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    grid = np.zeros((n, W, H, 3), np.uint8)
    agent = np.zeros((n, 8), np.int32)
    obst = np.zeros(n, np.uint64)
    flags = np.zeros(n, np.uint8)

    def run(mode):
        assert L.mg_selftest_dynobs(W, H, nob, s["start_x"], s["start_y"], s["start_dir"], redo, n, p(mode), p(grid), p(agent), p(words), p(obst), p(flags)) == 0

    orc.reset(seeds=seeds)
    run(np.full(n, 2, np.uint8))
    g2, a2 = orc.get_state()
    assert (grid == g2).all() and (agent[:, :3] == a2[:, :3]).all() and (words == orc.get_rng()).all() and not (flags & 1).any()
    rng = np.random.default_rng(5)
    pending = np.zeros(n, bool)
    n_resets = n_minus = 0
    for t in range(T):
        act = rng.integers(0, 7 if t % 5 == 0 else 3, n).astype(np.uint8)           # (actions >= 3 are "invalid": left)
        agent[:] = a2                                                                # the pose the moves must avoid / look ahead from
        run(np.where(pending, 2, 1).astype(np.uint8))
        _, rew, term, trunc, _, _ = orc.step(act)
        g2, a2 = orc.get_state()
        bad = np.argwhere((grid != g2).reshape(n, -1).any(1)).ravel()
        assert bad.size == 0, (env_id, t, bad[:4], pending[bad[:4]])
        assert (words == orc.get_rng()).all(), (env_id, t)
        assert (agent[pending, :3] == a2[pending, :3]).all()
        hit = (~pending) & (act == 2) & ((flags & 4) != 0)
        assert (rew[hit] == -1.0).all() and term[hit].all() and not (rew[(~pending) & ~hit] == -1.0).any()
        n_resets += int(pending.sum()); n_minus += int(hit.sum())
        pending = term | trunc
    assert n_resets > n // 2 and n_minus > 0


@pytest.mark.parametrize("env_id,n,T", [("MiniGrid-LavaCrossingS9N1-v0", 131, 50), ("MiniGrid-DoorKey-8x8-v0", 64, 60), ("MiniGrid-DoorKey-16x16-v0", 70, 40),
                                        ("BabyAI-GoToRedBall-v0", 200, 40), ("MiniGrid-Empty-5x5-v0", 64, 20)])
def test_full_observation_pipeline_on_the_host_equals_the_oracle(env_id, n, T):
    """mg_selftest_obs_full = k_roll7<., true>'s FullyObs observation (image-order code stream, agent cell, output-space encode) compiled for the
    host, against the oracle's FullyObsWrapper observation along random rollouts (doors in every state, carried objects gone from the grid,
    ragged last workgroups)."""
    import ctypes as C
    from oracle import oracle as O
    L = B.load()
    orc = O.OracleVec(env_id, n, full_obs=True)
    obs, _, _ = orc.reset(seeds=np.arange(n, dtype=np.uint64))
    rng = np.random.default_rng(2)
    out = np.zeros((n, orc.W, orc.H, 3), np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    for t in range(T):
        grid, agent = orc.get_state()
        assert L.mg_selftest_obs_full(orc.W, orc.H, n, p(np.ascontiguousarray(grid)), p(np.ascontiguousarray(agent)), p(out)) == 0
        bad = np.argwhere((out != obs).reshape(n, -1).any(1)).ravel()
        assert bad.size == 0, (env_id, t, bad[:5])
        a = rng.choice(7, size=n, p=[0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05]).astype(np.uint8)
        obs = orc.step(a)[0]
