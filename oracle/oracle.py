"""ctypes front-end of the CPU oracle (oracle/minigrid_oracle.c) — TEST INFRASTRUCTURE ONLY.

Never imported by the product package `minigrid_amd`.  Used by tests/ as the parity checker, by
`__graft_entry__.smoke()` as the checker, and by `bench.py`'s `cpu_baseline` leg as the timed CPU port.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# env kinds / object codes (mirror of the enums in minigrid_oracle.c)
K_EMPTY, K_DOORKEY, K_CROSSING, K_GOTO_REDBALL, K_LAVAGAP, K_DISTSHIFT, K_FOURROOMS, K_FETCH, K_GOTODOOR = 0, 1, 2, 3, 4, 5, 6, 7, 8
K_UNLOCK, K_UNLOCKPICKUP, K_BLOCKEDUNLOCKPICKUP, K_REDBLUEDOORS, K_MEMORY, K_KEYCORRIDOR = 9, 10, 11, 12, 13, 14
K_DYNOBS = 15
K_GOTO_REDBALLGREY, K_GOTO_REDBLUEBALL, K_GOTO_OBJ, K_GOTO_LOCAL, K_GOTOOBJECT = 16, 17, 18, 19, 20
K_LOCKEDROOM, K_PLAYGROUND, K_MULTIROOM = 21, 22, 23
K_PICKUPDIST, K_ONEROOM, K_OPENREDDOOR, K_PICKUPDIST_DEBUG, K_FINDOBJ = 24, 25, 26, 27, 28
K_UNLOCKLOCAL, K_BABYAI_KEYCORRIDOR, K_OBSTRUCTEDMAZE, K_PUTNEAR = 29, 30, 31, 32
K_BABYAI_GOTO, K_BABYAI_PICKUP, K_BABYAI_OPEN = 33, 34, 35
K_BABYAI_UNLOCKPICKUP, K_BABYAI_BLOCKEDUNLOCKPICKUP, K_UNLOCKTOUNLOCK, K_KEYINBOX, K_BABYAI_UNLOCK = 36, 37, 38, 39, 40
K_BABYAI_GOTODOOR, K_GOTOOBJDOOR, K_UNBLOCKPICKUP, K_PICKUPABOVE, K_GOTOIMPUNLOCK = 41, 42, 43, 44, 45
K_PUTNEXTLOCAL, K_PUTNEXT, K_ACTIONOBJDOOR, K_OPENDOOR = 46, 47, 48, 49
K_OPENTWODOORS, K_OPENDOORSORDER, K_MOVETWOACROSS, K_LEVELGEN = 50, 51, 52, 53
T_WALL, T_LAVA = 2, 9


class OracleCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "kind", "width", "height", "max_steps", "see_through", "start_x", "start_y", "start_dir",
        "num_crossings", "obstacle_type", "num_dists", "full_obs", "strip2_row", "view_size", "no_death_mask")] + [
        ("death_cost", C.c_double), ("room_size", C.c_int32), ("random_length", C.c_int32), ("done_actions", C.c_int32)]


OBS_KINDS = {"partial": 0, "full": 1, "onehot": 2, "symbolic": 3}
OBJECT_TO_IDX = {"unseen": 0, "empty": 1, "wall": 2, "floor": 3, "door": 4, "key": 5, "ball": 6, "box": 7, "goal": 8,
                 "lava": 9, "agent": 10}          # minigrid/core/constants.py:25-37


# Static config rows restated from the reference registry (minigrid/__init__.py:24-28,105-109,182-185,577-580
# and sibling rows) + constructor defaults (empty.py:68-91, doorkey.py:62-68, crossing.py:89-117,
# goto.py:129-131, roomgrid_level.py:77-83).
def spec(env_id: str) -> dict:
    def empty(size, random_start=False):
        return dict(kind=K_EMPTY, width=size, height=size, max_steps=4 * size * size, see_through=1,
                    start_x=-1 if random_start else 1, start_y=-1 if random_start else 1, start_dir=0,
                    missions=["get to the green goal square"])

    def doorkey(size):
        return dict(kind=K_DOORKEY, width=size, height=size, max_steps=10 * size * size, see_through=0,
                    missions=["use the key to open the door and then get to the goal"])

    def crossing(size, n, lava=True):
        return dict(kind=K_CROSSING, width=size, height=size, max_steps=4 * size * size, see_through=0,
                    num_crossings=n, obstacle_type=T_LAVA if lava else T_WALL,
                    missions=["avoid the lava and get to the green goal square" if lava
                              else "find the opening and get to the green goal square"])

    def lavagap(size):
        return dict(kind=K_LAVAGAP, width=size, height=size, max_steps=4 * size * size, see_through=0,
                    obstacle_type=T_LAVA, missions=["avoid the lava and get to the green goal square"])

    def distshift(row):
        return dict(kind=K_DISTSHIFT, width=9, height=7, max_steps=4 * 9 * 7, see_through=1, start_x=1, start_y=1,
                    start_dir=0, strip2_row=row, missions=["get to the green goal square"])

    color_names = ["blue", "green", "grey", "purple", "red", "yellow"]         # sorted COLOR_NAMES (constants.py:17)

    def fetch(size, n):
        # fetch.py:66-103: missions f"{syntax} {color} {type}", id = syntax*12 + color*2 + type
        syntax = ["get a", "go get a", "fetch a", "go fetch a", "you must fetch a"]
        return dict(kind=K_FETCH, width=size, height=size, max_steps=5 * size * size, see_through=1, num_dists=n,
                    missions=[f"{s} {c} {t}" for s in syntax for c in color_names for t in ("key", "ball")])

    def gotodoor(size):
        return dict(kind=K_GOTODOOR, width=size, height=size, max_steps=4 * size * size, see_through=1,
                    missions=[f"go to the {c} door" for c in color_names])

    def roomgrid(kind, room_size, rows, cols, max_steps, missions):
        # core/roomgrid.py:72-100: width = (room_size-1)*num_cols + 1, see_through_walls=False
        return dict(kind=kind, width=(room_size - 1) * cols + 1, height=(room_size - 1) * rows + 1, max_steps=max_steps,
                    see_through=0, room_size=room_size, missions=missions)

    def redblue(size):
        # redbluedoors.py:60-76: width = 2*size, max_steps = 20*size**2, default see_through_walls=False
        return dict(kind=K_REDBLUEDOORS, width=2 * size, height=size, max_steps=20 * size * size, see_through=0,
                    missions=["open the red door then the blue door"])

    def memory(size, random_length=False):
        # memory.py:69-90: max_steps = 5*size**2, see_through_walls=False
        return dict(kind=K_MEMORY, width=size, height=size, max_steps=5 * size * size, see_through=0,
                    random_length=int(random_length), missions=["go to the matching object at the end of the hallway"])

    def dynobs(size, n, random_start=False):
        # dynamicobstacles.py:72-106: n_obstacles clamped (:84-88), see_through_walls=True, max_steps = 4*size**2
        n = int(n) if n <= size / 2 + 1 else int(size / 2)
        return dict(kind=K_DYNOBS, width=size, height=size, max_steps=4 * size * size, see_through=1, num_dists=n,
                    start_x=-1 if random_start else 1, start_y=-1 if random_start else 1, start_dir=0,
                    missions=["get to the green goal square"])

    def babyai_goto(kind, room_size, num_dists, missions):
        # RoomGridLevel (roomgrid_level.py:60-85): 1x1 rooms, max_steps = num_navs(1) * room_size**2 per episode
        return dict(kind=kind, width=room_size, height=room_size, max_steps=room_size * room_size, see_through=0,
                    num_dists=num_dists, missions=missions)

    goto_obj_missions = [f"go to {art} {c} {t}" for art in ("the", "a") for c in color_names for t in ("key", "ball", "box")]

    def keycorridor(room_size, rows):
        # keycorridor.py:75-104: num_cols = 3 (RoomGrid default), max_steps = 30*room_size**2, obj_type "ball"
        return roomgrid(K_KEYCORRIDOR, room_size, rows, 3, 30 * room_size * room_size,
                        [f"pick up the {c} ball" for c in color_names])

    def gotoobject(size, n):
        # gotoobject.py:66-91: see_through_walls=True, max_steps = 5*size**2
        return dict(kind=K_GOTOOBJECT, width=size, height=size, max_steps=5 * size * size, see_through=1, num_dists=n,
                    missions=[f"go to the {c} {t}" for c in color_names for t in ("key", "ball", "box")])

    def multiroom(lo, hi, max_size):
        # multiroom.py:79-112: 25x25, max_steps = maxNumRooms * 20; rows minigrid/__init__.py:359-386
        return dict(kind=K_MULTIROOM, width=25, height=25, max_steps=hi * 20, see_through=0, num_crossings=lo, num_dists=hi,
                    room_size=max_size, missions=["traverse the rooms to get to the goal"])

    # "pick up " + ObjDesc.surface (verifier.py:73-103): article x (no colour | colour) x ("object" | type)
    pickup_missions = ["pick up " + art + " " + (c + " " if c else "") + t for art in ("the", "a")
                       for c in [""] + color_names for t in ("object", "key", "ball", "box")]

    def babyai_pickup(kind, room_size):
        # pickup.py:272-274 (PickupDist: room_size 7), other.py:326-327 (OneRoomS8 and room_size 12/16/20)
        return dict(kind=kind, width=room_size, height=room_size, max_steps=room_size * room_size, see_through=0,
                    room_size=room_size, missions=pickup_missions)

    def obstructedmaze(rows, cols, rooms_visited, key_in_box, blocked, num_quarters=1, agent_room=(0, 0), v1=False, one_d=False):
        # obstructedmaze.py:80-106: room_size 6, max_steps = 4 * num_rooms_visited * 36; flags in num_crossings (see the C file)
        return dict(kind=K_OBSTRUCTEDMAZE, width=cols * 5 + 1, height=rows * 5 + 1, max_steps=4 * rooms_visited * 36, see_through=0,
                    room_size=6, num_crossings=int(key_in_box) | int(blocked) << 1 | int(v1) << 2 | int(one_d) << 3,
                    num_dists=num_quarters, start_x=agent_room[0], start_y=agent_room[1], missions=["pick up the blue ball"])

    def levelgen(rs, rows, cols, num_dists, action_kinds, instr_kinds, locations, unblocking, implicit_unlock, locked_pct):
        return dict(kind=K_LEVELGEN, width=cols * (rs - 1) + 1, height=rows * (rs - 1) + 1, max_steps=rs * rs * rows * cols, see_through=0,
                    room_size=rs, num_dists=num_dists, strip2_row=locked_pct,
                    num_crossings=action_kinds | instr_kinds << 4 | int(locations) << 7 | int(unblocking) << 8 | int(implicit_unlock) << 9,
                    missions=[""])

    putnext_missions = [f"put the {c1} {t1} next to the {c2} {t2}" for c1 in color_names for t1 in ("key", "ball", "box")
                        for c2 in color_names for t2 in ("key", "ball", "box")]

    table = {
        # multi-room BabyAI levels with one instruction (oracle only so far): goto.py:403-426, pickup.py:66-72, open.py:69-86;
        # max_steps = 1 * room_size**2 * rows * cols (roomgrid_level.py:71-85); rows minigrid/__init__.py:681-731, 760-763, 848-851
        **{name: dict(kind=K_BABYAI_GOTO, width=cols * (rs - 1) + 1, height=rows * (rs - 1) + 1, max_steps=rs * rs * rows * cols,
                      see_through=0, room_size=rs, num_dists=nd, num_crossings=int(opened), missions=goto_obj_missions)
           for name, rs, rows, cols, nd, opened in (
               ("BabyAI-GoTo-v0", 8, 3, 3, 18, False), ("BabyAI-GoToOpen-v0", 8, 3, 3, 18, True),
               ("BabyAI-GoToObjMaze-v0", 8, 3, 3, 1, False), ("BabyAI-GoToObjMazeOpen-v0", 8, 3, 3, 1, True),
               ("BabyAI-GoToObjMazeS4R2-v0", 4, 2, 2, 1, False), ("BabyAI-GoToObjMazeS4-v0", 4, 3, 3, 1, False),
               ("BabyAI-GoToObjMazeS5-v0", 5, 3, 3, 1, False), ("BabyAI-GoToObjMazeS6-v0", 6, 3, 3, 1, False),
               ("BabyAI-GoToObjMazeS7-v0", 7, 3, 3, 1, False))},
        # unlock.py: UnlockPickup(-Dist) 1 x 2 rooms of 6, max_steps per episode = 1 * 36 * 2 (the `if max is None` slip at :299
        # leaves it to RoomGridLevel.reset); BlockedUnlockPickup 16 * 36; UnlockToUnlock 1 x 3 rooms, 30 * 36; KeyInBox / Unlock defaults
        "BabyAI-UnlockPickup-v0": dict(kind=K_BABYAI_UNLOCKPICKUP, width=11, height=6, max_steps=72, see_through=0, room_size=6, num_dists=0,
                                       missions=pickup_missions),
        "BabyAI-UnlockPickupDist-v0": dict(kind=K_BABYAI_UNLOCKPICKUP, width=11, height=6, max_steps=72, see_through=0, room_size=6,
                                           num_dists=4, missions=pickup_missions),
        "BabyAI-BlockedUnlockPickup-v0": dict(kind=K_BABYAI_BLOCKEDUNLOCKPICKUP, width=11, height=6, max_steps=576, see_through=0,
                                              room_size=6, missions=pickup_missions),
        "BabyAI-UnlockToUnlock-v0": dict(kind=K_UNLOCKTOUNLOCK, width=16, height=6, max_steps=1080, see_through=0, room_size=6,
                                         missions=pickup_missions),
        "BabyAI-KeyInBox-v0": dict(kind=K_KEYINBOX, width=22, height=22, max_steps=576, see_through=0, room_size=8, missions=["open the door"]),
        "BabyAI-Unlock-v0": dict(kind=K_BABYAI_UNLOCK, width=22, height=22, max_steps=576, see_through=0, room_size=8,
                                 missions=[f"open {art} {c} door" for art in ("the", "a") for c in color_names]),
        # goto.py:727-740 (room_size 7), :797-813, :486-531; pickup.py:128-140, :346-362 (room_size 6, max_steps 8 * 36)
        "BabyAI-GoToDoor-v0": dict(kind=K_BABYAI_GOTODOOR, width=19, height=19, max_steps=441, see_through=0, room_size=7,
                                   missions=[f"go to {art} {c} door" for art in ("the", "a") for c in color_names]),
        "BabyAI-GoToObjDoor-v0": dict(kind=K_GOTOOBJDOOR, width=22, height=22, max_steps=576, see_through=0, room_size=8,
                                      missions=[f"go to {art} {c} {t}" for art in ("the", "a") for c in color_names
                                                for t in ("key", "ball", "box", "door")]),
        "BabyAI-GoToImpUnlock-v0": dict(kind=K_GOTOIMPUNLOCK, width=22, height=22, max_steps=576, see_through=0, room_size=8,
                                        missions=goto_obj_missions),
        "BabyAI-UnblockPickup-v0": dict(kind=K_UNBLOCKPICKUP, width=22, height=22, max_steps=576, see_through=0, room_size=8,
                                        missions=pickup_missions),
        "BabyAI-PickupAbove-v0": dict(kind=K_PICKUPABOVE, width=16, height=16, max_steps=288, see_through=0, room_size=6,
                                      missions=pickup_missions),
        # putnext.py:68-80 (one room), :148-166 (1 x 2 rooms, max_steps 8 * room_size**2); other.py:83-106; open.py:203-229
        **{name: dict(kind=K_PUTNEXTLOCAL, width=rs, height=rs, max_steps=2 * rs * rs, see_through=0, room_size=rs, num_dists=n,
                      missions=putnext_missions)
           for name, rs, n in (("BabyAI-PutNextLocal-v0", 8, 8), ("BabyAI-PutNextLocalS5N3-v0", 5, 3), ("BabyAI-PutNextLocalS6N4-v0", 6, 4))},
        **{name: dict(kind=K_PUTNEXT, width=2 * (rs - 1) + 1, height=rs, max_steps=8 * rs * rs, see_through=0, room_size=rs, num_dists=n,
                      num_crossings=int(carrying), missions=putnext_missions)
           for name, rs, n, carrying in (("BabyAI-PutNextS4N1-v0", 4, 1, False), ("BabyAI-PutNextS5N2-v0", 5, 2, False),
                                         ("BabyAI-PutNextS5N1-v0", 5, 1, False), ("BabyAI-PutNextS6N3-v0", 6, 3, False),
                                         ("BabyAI-PutNextS7N4-v0", 7, 4, False), ("BabyAI-PutNextS5N2Carrying-v0", 5, 2, True),
                                         ("BabyAI-PutNextS6N3Carrying-v0", 6, 3, True), ("BabyAI-PutNextS7N4Carrying-v0", 7, 4, True))},
        "BabyAI-ActionObjDoor-v0": dict(kind=K_ACTIONOBJDOOR, width=19, height=19, max_steps=441, see_through=0, room_size=7,
                                        missions=[f"{verb} {art} {c} {t}" for verb in ("go to", "pick up", "open") for art in ("the", "a")
                                                  for c in color_names for t in ("key", "ball", "box", "door")]),
        **{name: dict(kind=K_OPENDOOR, width=22, height=22, max_steps=576, see_through=0, room_size=8, num_crossings=sel, strip2_row=int(dbg),
                      missions=[f"open the {c} door" for c in color_names] +
                               [f"open {art} door {loc}" for art in ("the", "a")
                                for loc in ("on your left", "on your right", "in front of you", "behind you")])
           for name, sel, dbg in (("BabyAI-OpenDoor-v0", 0, False), ("BabyAI-OpenDoorDebug-v0", 0, True),
                                  ("BabyAI-OpenDoorColor-v0", 1, False), ("BabyAI-OpenDoorLoc-v0", 2, False))},
        # open.py:289-325, :383-425 (room_size 6, max_steps 20 * 36); other.py:388-428 (1 x 2 rooms, max_steps 16 * room_size**2).
        # Their missions are sentences ("..., then ...", "... after you ..."): OracleVec.mission_strings()
        **{name: dict(kind=K_OPENTWODOORS, width=16, height=16, max_steps=720, see_through=0, room_size=6, start_x=c1, start_y=c2,
                      strip2_row=int(strict), missions=[""])
           for name, c1, c2, strict in (("BabyAI-OpenTwoDoors-v0", -1, -1, False), ("BabyAI-OpenRedBlueDoors-v0", 4, 0, False),
                                        ("BabyAI-OpenRedBlueDoorsDebug-v0", 4, 0, True))},
        **{name: dict(kind=K_OPENDOORSORDER, width=16, height=16, max_steps=720, see_through=0, room_size=6, num_dists=n,
                      strip2_row=int(dbg), missions=[""])
           for name, n, dbg in (("BabyAI-OpenDoorsOrderN2-v0", 2, False), ("BabyAI-OpenDoorsOrderN4-v0", 4, False),
                                ("BabyAI-OpenDoorsOrderN2Debug-v0", 2, True), ("BabyAI-OpenDoorsOrderN4Debug-v0", 4, True))},
        **{name: dict(kind=K_MOVETWOACROSS, width=2 * (rs - 1) + 1, height=rs, max_steps=16 * rs * rs, see_through=0, room_size=rs,
                      num_dists=n, missions=[""])
           for name, rs, n in (("BabyAI-MoveTwoAcrossS5N2-v0", 5, 2), ("BabyAI-MoveTwoAcrossS8N9-v0", 8, 9))},
        # LevelGen (levelgen.py:24-80) configurations: pickup.py:198-213, goto.py:590-606, synth.py:83-97, :168-178, :274-281,
        # :374-382, :476-480, :570-576.  max_steps is per episode (num_navs * room_size**2 * rooms); the value here is the 1-nav one
        **{name: levelgen(rs, rows, cols, nd, acts, kinds, loc, unb, imp, prob)
           for name, rs, rows, cols, nd, acts, kinds, loc, unb, imp, prob in (
               ("BabyAI-PickupLoc-v0", 8, 1, 1, 8, 0b0010, 0b001, True, False, True, 0),
               ("BabyAI-GoToSeq-v0", 8, 3, 3, 18, 0b0001, 0b111, False, False, True, 0),
               ("BabyAI-GoToSeqS5R2-v0", 5, 2, 2, 4, 0b0001, 0b111, False, False, True, 0),
               ("BabyAI-Synth-v0", 8, 3, 3, 18, 0b1111, 0b001, False, True, False, 50),
               ("BabyAI-SynthS5R2-v0", 5, 2, 3, 18, 0b1111, 0b001, False, True, False, 50),
               ("BabyAI-SynthLoc-v0", 8, 3, 3, 18, 0b1111, 0b001, True, True, False, 50),
               ("BabyAI-SynthSeq-v0", 8, 3, 3, 18, 0b1111, 0b111, True, True, False, 50),
               ("BabyAI-MiniBossLevel-v0", 5, 2, 2, 7, 0b1111, 0b111, True, True, True, 25),
               ("BabyAI-BossLevel-v0", 8, 3, 3, 18, 0b1111, 0b111, True, True, True, 50),
               ("BabyAI-BossLevelNoUnlock-v0", 8, 3, 3, 18, 0b1111, 0b111, True, True, False, 0))},
        "BabyAI-Pickup-v0": dict(kind=K_BABYAI_PICKUP, width=22, height=22, max_steps=576, see_through=0, room_size=8, num_dists=18,
                                 missions=pickup_missions),
        "BabyAI-Open-v0": dict(kind=K_BABYAI_OPEN, width=22, height=22, max_steps=576, see_through=0, room_size=8, num_dists=18,
                               missions=[f"open {art} {c} door" for art in ("the", "a") for c in color_names]),
        # putnear.py:68-93: see_through_walls=True, max_steps = 5 * size; rows minigrid/__init__.py:526-537 (oracle only so far)
        **{name: dict(kind=K_PUTNEAR, width=size, height=size, max_steps=5 * size, see_through=1, num_dists=n,
                      missions=[f"put the {mc} {mt} near the {tc} {tt}" for mc in color_names for mt in ("key", "ball", "box")
                                for tc in color_names for tt in ("key", "ball", "box")])
           for name, size, n in (("MiniGrid-PutNear-6x6-N2-v0", 6, 2), ("MiniGrid-PutNear-8x8-N3-v0", 8, 3))},
        # rows minigrid/__init__.py:390-515 (NOT yet on the device: oracle groundwork for the next widening step)
        "MiniGrid-ObstructedMaze-1Dl-v0": obstructedmaze(1, 2, 2, False, False, one_d=True),
        "MiniGrid-ObstructedMaze-1Dlh-v0": obstructedmaze(1, 2, 2, True, False, one_d=True),
        "MiniGrid-ObstructedMaze-1Dlhb-v0": obstructedmaze(1, 2, 2, True, True, one_d=True),
        "MiniGrid-ObstructedMaze-2Dl-v0": obstructedmaze(3, 3, 4, False, False, 1, (2, 1)),
        "MiniGrid-ObstructedMaze-2Dlh-v0": obstructedmaze(3, 3, 4, True, False, 1, (2, 1)),
        "MiniGrid-ObstructedMaze-2Dlhb-v0": obstructedmaze(3, 3, 4, True, True, 1, (2, 1)),
        "MiniGrid-ObstructedMaze-1Q-v0": obstructedmaze(3, 3, 5, True, True, 1, (1, 1)),
        "MiniGrid-ObstructedMaze-2Q-v0": obstructedmaze(3, 3, 11, True, True, 2, (2, 1)),
        "MiniGrid-ObstructedMaze-Full-v0": obstructedmaze(3, 3, 25, True, True, 4, (1, 1)),
        "MiniGrid-ObstructedMaze-2Dlhb-v1": obstructedmaze(3, 3, 4, True, True, 1, (2, 1), v1=True),
        "MiniGrid-ObstructedMaze-1Q-v1": obstructedmaze(3, 3, 5, True, True, 1, (1, 1), v1=True),
        "MiniGrid-ObstructedMaze-2Q-v1": obstructedmaze(3, 3, 11, True, True, 2, (2, 1), v1=True),
        "MiniGrid-ObstructedMaze-Full-v1": obstructedmaze(3, 3, 25, True, True, 4, (1, 1), v1=True),
        # other.py:163-167: 3 x 3 rooms, max_steps = 20 * room_size**2 (fixed)
        **{f"BabyAI-FindObjS{rs}-v0": dict(kind=K_FINDOBJ, width=3 * (rs - 1) + 1, height=3 * (rs - 1) + 1, max_steps=20 * rs * rs,
                                          see_through=0, room_size=rs, missions=pickup_missions) for rs in (5, 6, 7)},
        # unlock.py:163-174: default RoomGridLevel geometry (3 x 3 rooms of size 8), max_steps = 1 * 64 * 9
        "BabyAI-UnlockLocal-v0": dict(kind=K_UNLOCKLOCAL, width=22, height=22, max_steps=576, see_through=0, room_size=8, num_dists=0,
                                      missions=["open the door"]),
        "BabyAI-UnlockLocalDist-v0": dict(kind=K_UNLOCKLOCAL, width=22, height=22, max_steps=576, see_through=0, room_size=8,
                                          num_dists=3, missions=["open the door"]),
        # other.py:231-250: 3 columns x num_rows rooms, max_steps = 30 * room_size**2
        **{name: dict(kind=K_BABYAI_KEYCORRIDOR, width=3 * (rs - 1) + 1, height=rows * (rs - 1) + 1, max_steps=30 * rs * rs,
                      see_through=0, room_size=rs, missions=pickup_missions)
           for name, rs, rows in (("BabyAI-KeyCorridor-v0", 6, 3), ("BabyAI-KeyCorridorS3R1-v0", 3, 1), ("BabyAI-KeyCorridorS3R2-v0", 3, 2),
                                  ("BabyAI-KeyCorridorS3R3-v0", 3, 3), ("BabyAI-KeyCorridorS4R3-v0", 4, 3),
                                  ("BabyAI-KeyCorridorS5R3-v0", 5, 3), ("BabyAI-KeyCorridorS6R3-v0", 6, 3))},
        "BabyAI-PickupDist-v0": babyai_pickup(K_PICKUPDIST, 7), "BabyAI-PickupDistDebug-v0": babyai_pickup(K_PICKUPDIST_DEBUG, 7),
        "BabyAI-OneRoomS8-v0": babyai_pickup(K_ONEROOM, 8), "BabyAI-OneRoomS12-v0": babyai_pickup(K_ONEROOM, 12),
        "BabyAI-OneRoomS16-v0": babyai_pickup(K_ONEROOM, 16), "BabyAI-OneRoomS20-v0": babyai_pickup(K_ONEROOM, 20),
        # open.py:140-146: 1 x 2 rooms of size 5 -> 9 x 5 grid, max_steps = 1 * 25 * 2
        "BabyAI-OpenRedDoor-v0": dict(kind=K_OPENREDDOOR, width=9, height=5, max_steps=50, see_through=0, room_size=5,
                                      missions=["open the red door"]),
        # lockedroom.py:82-102: size 19, max_steps = 10*size; playground.py:16-25: 19x19, max_steps 100
        "MiniGrid-LockedRoom-v0": dict(kind=K_LOCKEDROOM, width=19, height=19, max_steps=190, see_through=0,
                                       missions=[f"get the {a} key from the {b} room, unlock the {a} door and go to the goal"
                                                 for a in color_names for b in color_names]),
        "MiniGrid-Playground-v0": dict(kind=K_PLAYGROUND, width=19, height=19, max_steps=100, see_through=0, missions=[""]),
        "MiniGrid-MultiRoom-N2-S4-v0": multiroom(2, 2, 4), "MiniGrid-MultiRoom-N4-S5-v0": multiroom(6, 6, 5),
        "MiniGrid-MultiRoom-N4-S5-v1": multiroom(4, 4, 5), "MiniGrid-MultiRoom-N6-v0": multiroom(6, 6, 10),
        "MiniGrid-GoToObject-6x6-N2-v0": gotoobject(6, 2), "MiniGrid-GoToObject-8x8-N2-v0": gotoobject(8, 2),
        # envs/babyai/goto.py: GoToRedBallGrey :63-78, GoToRedBlueBall :657-677, GoToObj :253-260, GoToLocal :329-338;
        # registry rows minigrid/__init__.py:572-679, 750-753
        "BabyAI-GoToRedBallGrey-v0": babyai_goto(K_GOTO_REDBALLGREY, 8, 7, ["go to the red ball", "go to a red ball"]),
        "BabyAI-GoToRedBlueBall-v0": babyai_goto(K_GOTO_REDBLUEBALL, 8, 7, ["go to the red ball", "go to the blue ball"]),
        "BabyAI-GoToObj-v0": babyai_goto(K_GOTO_OBJ, 8, 1, goto_obj_missions),
        "BabyAI-GoToObjS4-v0": babyai_goto(K_GOTO_OBJ, 4, 1, goto_obj_missions),
        "BabyAI-GoToObjS6-v1": babyai_goto(K_GOTO_OBJ, 6, 1, goto_obj_missions),
        "BabyAI-GoToLocal-v0": babyai_goto(K_GOTO_LOCAL, 8, 8, goto_obj_missions),
        **{f"BabyAI-GoToLocalS{s_}N{n_}-v0": babyai_goto(K_GOTO_LOCAL, s_, n_, goto_obj_missions)
           for s_, n_ in ((5, 2), (6, 2), (6, 3), (6, 4), (7, 4), (7, 5), (8, 2), (8, 3), (8, 4), (8, 5), (8, 6), (8, 7))},
        "MiniGrid-Dynamic-Obstacles-5x5-v0": dynobs(5, 2), "MiniGrid-Dynamic-Obstacles-Random-5x5-v0": dynobs(5, 2, True),
        "MiniGrid-Dynamic-Obstacles-6x6-v0": dynobs(6, 3), "MiniGrid-Dynamic-Obstacles-Random-6x6-v0": dynobs(6, 3, True),
        "MiniGrid-Dynamic-Obstacles-8x8-v0": dynobs(8, 4), "MiniGrid-Dynamic-Obstacles-16x16-v0": dynobs(16, 8),
        **{f"MiniGrid-KeyCorridorS{s_}R{r_}-v0": keycorridor(s_, r_) for s_, r_ in ((3, 1), (3, 2), (3, 3), (4, 3), (5, 3), (6, 3))},
        "MiniGrid-RedBlueDoors-6x6-v0": redblue(6), "MiniGrid-RedBlueDoors-8x8-v0": redblue(8),
        "MiniGrid-MemoryS17Random-v0": memory(17, True), "MiniGrid-MemoryS13Random-v0": memory(13, True),
        "MiniGrid-MemoryS13-v0": memory(13), "MiniGrid-MemoryS11-v0": memory(11), "MiniGrid-MemoryS9-v0": memory(9),
        "MiniGrid-MemoryS7-v0": memory(7),
        # unlock.py:52-70, unlockpickup.py:57-80, blockedunlockpickup.py:65-88 (room_size 6, 1x2 rooms)
        "MiniGrid-Unlock-v0": roomgrid(K_UNLOCK, 6, 1, 2, 8 * 36, ["open the door"]),
        "MiniGrid-UnlockPickup-v0": roomgrid(K_UNLOCKPICKUP, 6, 1, 2, 8 * 36, [f"pick up the {c} box" for c in color_names]),
        "MiniGrid-BlockedUnlockPickup-v0": roomgrid(K_BLOCKEDUNLOCKPICKUP, 6, 1, 2, 16 * 36,
                                                   [f"pick up the {c} {t}" for c in color_names for t in ("box", "key")]),
        "MiniGrid-Fetch-5x5-N2-v0": fetch(5, 2), "MiniGrid-Fetch-6x6-N2-v0": fetch(6, 2), "MiniGrid-Fetch-8x8-N3-v0": fetch(8, 3),
        "MiniGrid-GoToDoor-5x5-v0": gotodoor(5), "MiniGrid-GoToDoor-6x6-v0": gotodoor(6), "MiniGrid-GoToDoor-8x8-v0": gotodoor(8),
        # lavagap.py:68-91, distshift.py:65-93, fourrooms.py:59-73 + their registry rows (minigrid/__init__.py:78-88,213-216,294-310)
        "MiniGrid-LavaGapS5-v0": lavagap(5), "MiniGrid-LavaGapS6-v0": lavagap(6), "MiniGrid-LavaGapS7-v0": lavagap(7),
        "MiniGrid-DistShift1-v0": distshift(2), "MiniGrid-DistShift2-v0": distshift(5),
        "MiniGrid-FourRooms-v0": dict(kind=K_FOURROOMS, width=19, height=19, max_steps=100, see_through=0,
                                      missions=["reach the goal"]),
        "MiniGrid-Empty-5x5-v0": empty(5), "MiniGrid-Empty-Random-5x5-v0": empty(5, True),
        "MiniGrid-Empty-6x6-v0": empty(6), "MiniGrid-Empty-Random-6x6-v0": empty(6, True),
        "MiniGrid-Empty-8x8-v0": empty(8), "MiniGrid-Empty-16x16-v0": empty(16),
        "MiniGrid-DoorKey-5x5-v0": doorkey(5), "MiniGrid-DoorKey-6x6-v0": doorkey(6),
        "MiniGrid-DoorKey-8x8-v0": doorkey(8), "MiniGrid-DoorKey-16x16-v0": doorkey(16),
        "MiniGrid-LavaCrossingS9N1-v0": crossing(9, 1), "MiniGrid-LavaCrossingS9N2-v0": crossing(9, 2),
        "MiniGrid-LavaCrossingS9N3-v0": crossing(9, 3), "MiniGrid-LavaCrossingS11N5-v0": crossing(11, 5),
        "MiniGrid-SimpleCrossingS9N1-v0": crossing(9, 1, False), "MiniGrid-SimpleCrossingS9N2-v0": crossing(9, 2, False),
        "MiniGrid-SimpleCrossingS9N3-v0": crossing(9, 3, False), "MiniGrid-SimpleCrossingS11N5-v0": crossing(11, 5, False),
        "BabyAI-GoToRedBall-v0": dict(kind=K_GOTO_REDBALL, width=8, height=8, max_steps=64, see_through=0, num_dists=7,
                                      missions=["go to the red ball", "go to a red ball"]),
        "BabyAI-GoToRedBallNoDists-v0": dict(kind=K_GOTO_REDBALL, width=8, height=8, max_steps=64, see_through=0,
                                             num_dists=0, missions=["go to the red ball", "go to a red ball"]),
    }
    return table[env_id]


def build(force: bool = False) -> str:
    """Compile oracle/minigrid_oracle.c -> oracle/liboracle.so (gcc; seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "minigrid_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(OracleCfg), C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_obs_bytes.argtypes = [C.c_void_p]
        vp = C.c_void_p
        L.oracle_reset.argtypes = [vp, vp, vp, vp, vp, vp]
        L.oracle_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
        for f in (L.oracle_get_state, L.oracle_set_state):
            f.argtypes = [vp, vp, vp]
        for f in (L.oracle_get_rng, L.oracle_set_rng, L.oracle_get_missions, L.oracle_get_mission_strs, L.oracle_get_stuck):
            f.argtypes = [vp, vp]
        L.oracle_rng_kat.argtypes = [C.c_uint64, vp, vp, C.c_int, vp, C.c_int, C.c_int64]
        L.oracle_shuffle_kat.argtypes = [C.c_uint64, vp, C.c_int]
        L.oracle_reward_lut.argtypes = [C.c_int, vp]
        L.oracle_philox_actions.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_int, vp]
        L.oracle_philox4x32_10.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.oracle_rollout.restype = C.c_uint64
        L.oracle_rollout.argtypes = [vp, C.c_int, C.c_uint64, vp]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleVec:
    """N independent reference-semantics envs stepped in lockstep on the CPU (scalar C)."""

    def __init__(self, env_id: str, num_envs: int, full_obs: bool = False, obs: str | None = None, view_size: int = 7,
                 no_death_types=(), death_cost: float = -1.0, tile_size: int = 8, highlight: bool = True, done_actions=False,
                 **overrides):
        """obs: "partial" | "full" (FullyObsWrapper) | "onehot" (OneHotPartialObsWrapper) | "symbolic"
        (SymbolicObsWrapper, returned as int8); view_size: ViewSizeWrapper; no_death_types/death_cost: NoDeath."""
        s = dict(spec(env_id))
        s.update(overrides)
        self.missions = s.pop("missions")
        # "rgb" (RGBImgObsWrapper) / "rgb_partial" (RGBImgPartialObsWrapper): the C core steps with its partial
        # observation, oracle/render.py draws the frame from the state
        self.rgb = obs if obs in ("rgb", "rgb_partial") else None
        self.tile_size, self.highlight = int(tile_size), bool(highlight)
        if self.rgb:
            obs = "partial"
        kind = OBS_KINDS[obs] if obs is not None else int(bool(full_obs))
        mask = 0
        for t in no_death_types:
            mask |= 1 << OBJECT_TO_IDX[t]
        self.cfg = OracleCfg(full_obs=kind, view_size=int(view_size), no_death_mask=mask, death_cost=float(death_cost),
                             done_actions=2 if done_actions == "enum" else int(bool(done_actions)),      # BABYAI_DONE_ACTIONS (verifier.py:26); "enum": stepped with Actions members (:561)
                             **{k: int(v) for k, v in s.items()})
        self.n = num_envs
        self.W, self.H = self.cfg.width, self.cfg.height
        self.full_obs = kind == 1
        self.h = lib().oracle_create(C.byref(self.cfg), num_envs)
        if not self.h:
            raise RuntimeError("oracle_create failed")
        self.obs_shape = {0: (view_size, view_size, 3), 1: (self.W, self.H, 3), 2: (view_size, view_size, 20),
                          3: (self.W, self.H, 3)}[kind]
        self.obs_dtype = np.int8 if kind == 3 else np.uint8

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_destroy(self.h)
            self.h = None

    def _outs(self):
        n = self.n
        return (np.zeros((n,) + self.obs_shape, self.obs_dtype), np.zeros(n, np.uint8), np.zeros(n, np.uint8))

    def _missions(self, m):
        """Mission ids: the C core reports a byte per env; levels with more than 256 missions are re-read as int32."""
        if len(self.missions) <= 256:
            return m
        out = np.zeros(self.n, np.int32)
        lib().oracle_get_missions(self.h, _p(out))
        return out.astype(np.uint16)

    def mission_strings(self):
        """Instr.surface() per env, for the levels whose missions are sentences (kinds on the general verifier)."""
        buf = np.zeros((self.n, 256), np.uint8)
        lib().oracle_get_mission_strs(self.h, _p(buf))
        return np.array([bytes(r).split(b"\0", 1)[0].decode() for r in buf])

    def _frame(self, obs):
        if not self.rgb:
            return obs
        from . import render
        if self.rgb == "rgb":
            grid, agent = self.get_state()
            return render.render_full(grid, agent, obs, self.tile_size, self.highlight)
        return render.render_pov(obs, self.tile_size)

    def reset(self, seeds=None, mask=None):
        obs, d, m = self._outs()
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().oracle_reset(self.h, _p(sd), _p(mk), _p(obs), _p(d), _p(m))
        return self._frame(obs), d, self._missions(m)

    def step_quiet(self, actions, autoreset: int = 1):
        """step() without producing the observation (replaying a long rollout whose observations are compared on a sample of
        steps only): returns (reward, terminated, truncated)."""
        a = np.ascontiguousarray(actions, dtype=np.uint8)
        rew = np.zeros(self.n, np.float64)
        term = np.zeros(self.n, np.uint8)
        trunc = np.zeros(self.n, np.uint8)
        rc = lib().oracle_step(self.h, _p(a), autoreset, None, _p(rew), _p(term), _p(trunc), None, None)
        if rc == -1:
            raise ValueError("Unknown action")
        if rc:
            raise AssertionError("front cell out of bounds")
        return rew, term.astype(bool), trunc.astype(bool)

    def step(self, actions, autoreset: int = 1):
        obs, d, m = self._outs()
        a = np.ascontiguousarray(actions, dtype=np.uint8)
        rew = np.zeros(self.n, np.float64)
        term = np.zeros(self.n, np.uint8)
        trunc = np.zeros(self.n, np.uint8)
        rc = lib().oracle_step(self.h, _p(a), autoreset, _p(obs), _p(rew), _p(term), _p(trunc), _p(d), _p(m))
        if rc == -1:
            raise ValueError("Unknown action")
        if rc:
            raise AssertionError("front cell out of bounds")
        return self._frame(obs), rew, term.astype(bool), trunc.astype(bool), d, self._missions(m)

    def get_state(self):
        grid = np.zeros((self.n, self.W, self.H, 3), np.uint8)
        agent = np.zeros((self.n, 8), np.int32)
        lib().oracle_get_state(self.h, _p(grid), _p(agent))
        return grid, agent

    def set_state(self, grid, agent):
        grid = np.ascontiguousarray(grid, np.uint8)
        agent = np.ascontiguousarray(agent, np.int32)
        assert grid.shape == (self.n, self.W, self.H, 3) and agent.shape == (self.n, 8)
        lib().oracle_set_state(self.h, _p(grid), _p(agent))

    def stuck(self):
        """Per env: the current episode's generation met RoomGrid.place_agent's endless loop (the reference would hang)."""
        r = np.zeros(self.n, np.uint8)
        lib().oracle_get_stuck(self.h, _p(r))
        return r.astype(bool)

    def get_rng(self):
        r = np.zeros((self.n, 5), np.uint64)
        lib().oracle_get_rng(self.h, _p(r))
        return r

    def set_rng(self, r):
        r = np.ascontiguousarray(r, np.uint64)
        lib().oracle_set_rng(self.h, _p(r))

    def rollout(self, T: int, action_seed: int = 0) -> int:
        scratch = np.zeros(int(np.prod(self.obs_shape)), np.uint8)
        return int(lib().oracle_rollout(self.h, T, action_seed, _p(scratch)))


def philox_actions(action_seed: int, t: int, n: int, env_base: int = 0):
    """Actions of step counter `t` of the product's device policy (mg_rollout) for envs env_base .. env_base + n - 1."""
    out = np.zeros(n, np.uint8)
    lib().oracle_philox_actions(int(action_seed), int(t), int(env_base), int(n), _p(out))
    return out


def philox4x32_10(counter, key):
    c = np.asarray(counter, np.uint32).copy()
    lib().oracle_philox4x32_10(_p(c), int(key[0]), int(key[1]))
    return c


def rng_kat(seed: int, n32: int = 16, nb: int = 16, bound_hi: int = 7):
    ss = np.zeros(4, np.uint64)
    n32o = np.zeros(n32, np.uint32)
    bo = np.zeros(nb, np.int64)
    lib().oracle_rng_kat(seed, _p(ss), _p(n32o), n32, _p(bo), nb, bound_hi)
    return ss, n32o, bo


def shuffle_kat(seed: int, n: int):
    a = np.arange(n, dtype=np.int32)
    lib().oracle_shuffle_kat(seed, _p(a), n)
    return a


def reward_lut(max_steps: int):
    out = np.zeros(max_steps + 1, np.float64)
    lib().oracle_reward_lut(max_steps, _p(out))
    return out
