"""Static config table: env id -> mg_config fields.

These rows restate the reference registry (`minigrid/__init__.py`: e.g. :24-28 LavaCrossingS9N1, :105-109 DoorKey-8x8,
:182-185 Empty-8x8, :577-580 BabyAI-GoToRedBall) together with the constructor defaults of each env class
(`envs/empty.py:68-91`, `envs/doorkey.py:62-68`, `envs/crossing.py:89-117`, `envs/babyai/goto.py:129-131`,
`envs/babyai/core/roomgrid_level.py:77-83` for the per-episode max_steps).  It is the reference's "plugin API"
(string id -> class + kwargs) reduced to data.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, Tuple

# mirror of include/minigrid_hip.h enums
ENV_EMPTY, ENV_DOORKEY, ENV_CROSSING, ENV_GOTO_REDBALL, ENV_LAVAGAP, ENV_DISTSHIFT, ENV_FOURROOMS, ENV_FETCH, ENV_GOTODOOR = 0, 1, 2, 3, 4, 5, 6, 7, 8
ENV_UNLOCK, ENV_UNLOCKPICKUP, ENV_BLOCKEDUNLOCKPICKUP, ENV_REDBLUEDOORS, ENV_MEMORY, ENV_KEYCORRIDOR = 9, 10, 11, 12, 13, 14
ENV_DYNOBS = 15
ENV_GOTO_REDBALLGREY, ENV_GOTO_REDBLUEBALL, ENV_GOTO_OBJ, ENV_GOTO_LOCAL, ENV_GOTOOBJECT = 16, 17, 18, 19, 20
ENV_LOCKEDROOM, ENV_PLAYGROUND, ENV_MULTIROOM = 21, 22, 23
ENV_PICKUPDIST, ENV_ONEROOM, ENV_OPENREDDOOR, ENV_PICKUPDIST_DEBUG, ENV_FINDOBJ = 24, 25, 26, 27, 28
ENV_UNLOCKLOCAL, ENV_BABYAI_KEYCORRIDOR, ENV_OBSTRUCTEDMAZE, ENV_PUTNEAR = 29, 30, 31, 32
ENV_BABYAI_GOTO, ENV_BABYAI_PICKUP, ENV_BABYAI_OPEN = 33, 34, 35
ENV_BABYAI_UNLOCKPICKUP, ENV_BABYAI_BLOCKEDUNLOCKPICKUP, ENV_UNLOCKTOUNLOCK, ENV_BABYAI_UNLOCK = 36, 37, 38, 40
ENV_KEYINBOX = 39
ENV_BABYAI_GOTODOOR, ENV_GOTOOBJDOOR, ENV_UNBLOCKPICKUP, ENV_PICKUPABOVE, ENV_GOTOIMPUNLOCK = 41, 42, 43, 44, 45
ENV_PUTNEXTLOCAL, ENV_PUTNEXT, ENV_ACTIONOBJDOOR, ENV_OPENDOOR = 46, 47, 48, 49
ENV_OPENTWODOORS, ENV_OPENDOORSORDER, ENV_MOVETWOACROSS, ENV_LEVELGEN = 50, 51, 52, 53      # the sentence levels
OBJ_WALL, OBJ_LAVA = 2, 9


@dataclass(frozen=True)
class EnvSpec:
    id: str
    env_kind: int
    width: int
    height: int
    max_steps: int
    see_through_walls: bool
    missions: Tuple[str, ...]
    agent_start: Tuple[int, int, int] = (-1, -1, 0)   # (x, y, dir); x < 0 => place_agent()
    num_crossings: int = 0
    obstacle_type: int = OBJ_LAVA
    num_dists: int = 0
    strip2_row: int = 0
    room_size: int = 0
    random_length: bool = False
    entry_point: str = ""                              # the reference class this row configures
    kwargs: dict = field(default_factory=dict)

    def with_max_steps(self, max_steps: int) -> "EnvSpec":
        return replace(self, max_steps=int(max_steps))


def _empty(id_, size, random_start=False):
    return EnvSpec(id_, ENV_EMPTY, size, size, 4 * size * size, True, ("get to the green goal square",),
                   agent_start=(-1, -1, 0) if random_start else (1, 1, 0), entry_point="minigrid.envs:EmptyEnv",
                   kwargs={"size": size, **({"agent_start_pos": None} if random_start else {})})


def _obstructedmaze(name, rows, cols, rooms_visited, key_in_box, blocked, num_quarters, agent_room, v1, one_d):
    flags = int(key_in_box) | int(blocked) << 1 | int(v1) << 2 | int(one_d) << 3
    if one_d:       # minigrid/__init__.py:390-406: class ObstructedMaze_1Dlhb; the registered kwargs, 1Dlhb relies on the defaults
        kw = {} if (key_in_box and blocked) else {"key_in_box": key_in_box, "blocked": blocked}
        ep = "minigrid.envs:ObstructedMaze_1Dlhb"
    else:           # :408-515; the Full rows rely on the class defaults
        kw = {} if name.startswith("Full") else {"agent_room": agent_room, "key_in_box": key_in_box, "blocked": blocked,
                                                 "num_quarters": num_quarters, "num_rooms_visited": rooms_visited}
        ep = "minigrid.envs:ObstructedMaze_Full_V1" if v1 else "minigrid.envs:ObstructedMaze_Full"
    return EnvSpec(f"MiniGrid-ObstructedMaze-{name}", ENV_OBSTRUCTEDMAZE, cols * 5 + 1, rows * 5 + 1, 4 * rooms_visited * 36, False,
                   ("pick up the blue ball",), agent_start=(agent_room[0], agent_room[1], 0), num_crossings=flags,
                   num_dists=num_quarters, room_size=6, entry_point=ep, kwargs=kw)


def _doorkey(id_, size):
    return EnvSpec(id_, ENV_DOORKEY, size, size, 10 * size * size, False,
                   ("use the key to open the door and then get to the goal",),
                   entry_point="minigrid.envs:DoorKeyEnv", kwargs={"size": size})


def _crossing(id_, size, n, lava=True):
    return EnvSpec(id_, ENV_CROSSING, size, size, 4 * size * size, False,
                   ("avoid the lava and get to the green goal square" if lava
                    else "find the opening and get to the green goal square",),
                   num_crossings=n, obstacle_type=OBJ_LAVA if lava else OBJ_WALL,
                   entry_point="minigrid.envs:CrossingEnv",
                   kwargs={"size": size, "num_crossings": n, **({} if lava else {"obstacle_type": "wall"})})


def _goto_red_ball(id_, num_dists):
    # room_size 8, 1x1 rooms => 8x8 grid; max_steps = num_navs(1) * room_size**2 * rows * cols = 64
    return EnvSpec(id_, ENV_GOTO_REDBALL, 8, 8, 64, False, ("go to the red ball", "go to a red ball"),
                   num_dists=num_dists, entry_point="minigrid.envs.babyai:GoToRedBall",
                   kwargs={} if num_dists == 7 else {"num_dists": num_dists})


def _lavagap(id_, size):
    # envs/lavagap.py:68-91 (obstacle_type=Lava, max_steps = 4*size**2), registry rows minigrid/__init__.py:294-310
    return EnvSpec(id_, ENV_LAVAGAP, size, size, 4 * size * size, False,
                   ("avoid the lava and get to the green goal square",), agent_start=(1, 1, 0), obstacle_type=OBJ_LAVA,
                   entry_point="minigrid.envs:LavaGapEnv", kwargs={"size": size})


def _distshift(id_, strip2_row):
    # envs/distshift.py:65-93 (9x7, see_through_walls=True, max_steps = 4*w*h), rows minigrid/__init__.py:78-88
    return EnvSpec(id_, ENV_DISTSHIFT, 9, 7, 4 * 9 * 7, True, ("get to the green goal square",), agent_start=(1, 1, 0),
                   strip2_row=strip2_row, entry_point="minigrid.envs:DistShiftEnv", kwargs={"strip2_row": strip2_row})


_COLOR_NAMES = ("blue", "green", "grey", "purple", "red", "yellow")       # sorted, core/constants.py:17


def _fetch(id_, size, num_objs):
    # envs/fetch.py:66-103: see_through_walls=True, max_steps = 5*size**2; mission id = syntax*12 + colour*2 + type,
    # the order of the MissionSpace's ordered placeholders.  Rows minigrid/__init__.py:196-208
    syntax = ("get a", "go get a", "fetch a", "go fetch a", "you must fetch a")
    missions = tuple(f"{s} {c} {t}" for s in syntax for c in _COLOR_NAMES for t in ("key", "ball"))
    return EnvSpec(id_, ENV_FETCH, size, size, 5 * size * size, True, missions, num_dists=num_objs,
                   entry_point="minigrid.envs:FetchEnv", kwargs={"size": size, "numObjs": num_objs})


def _gotodoor(id_, size):
    # envs/gotodoor.py:66-86: see_through_walls=True, max_steps = 4*size**2.  Rows minigrid/__init__.py:221-236
    return EnvSpec(id_, ENV_GOTODOOR, size, size, 4 * size * size, True,
                   tuple(f"go to the {c} door" for c in _COLOR_NAMES),
                   entry_point="minigrid.envs:GoToDoorEnv", kwargs={"size": size})


def _roomgrid_1x2(id_, kind, room_size, max_steps, missions, entry_point):
    # core/roomgrid.py:72-100: width = (room_size-1)*num_cols + 1, height = (room_size-1)*num_rows + 1, see_through_walls=False
    return EnvSpec(id_, kind, (room_size - 1) * 2 + 1, room_size, max_steps, False, missions, room_size=room_size,
                   entry_point=entry_point)


def _dynobs(id_, size, n_obstacles, random_start=False):
    # envs/dynamicobstacles.py:72-106: n_obstacles clamped (:84-88), see_through_walls=True, max_steps = 4*size**2;
    # rows minigrid/__init__.py:120-153.  Actions >= 3 count as 0 ("Invalid action", :137-139): no ValueError here.
    n = int(n_obstacles) if n_obstacles <= size / 2 + 1 else int(size / 2)
    return EnvSpec(id_, ENV_DYNOBS, size, size, 4 * size * size, True, ("get to the green goal square",),
                   agent_start=(-1, -1, 0) if random_start else (1, 1, 0), num_dists=n,
                   entry_point="minigrid.envs:DynamicObstaclesEnv",
                   kwargs={"size": size, "n_obstacles": n_obstacles, **({"agent_start_pos": None} if random_start else {})})


def _gotoobject(id_, size, n):
    # envs/gotoobject.py:66-91: see_through_walls=True, max_steps = 5*size**2; rows minigrid/__init__.py:238-251
    return EnvSpec(id_, ENV_GOTOOBJECT, size, size, 5 * size * size, True,
                   tuple(f"go to the {c} {t}" for c in _COLOR_NAMES for t in ("key", "ball", "box")), num_dists=n,
                   entry_point="minigrid.envs:GoToObjectEnv", kwargs={"size": size, "numObjs": n})


def _multiroom(id_, lo, hi, max_size=10):
    # envs/multiroom.py:79-112: 25x25, max_steps = maxNumRooms * 20; rows minigrid/__init__.py:359-386
    kw = {"minNumRooms": lo, "maxNumRooms": hi}
    if max_size != 10:
        kw["maxRoomSize"] = max_size
    return EnvSpec(id_, ENV_MULTIROOM, 25, 25, hi * 20, False, ("traverse the rooms to get to the goal",),
                   num_crossings=lo, num_dists=hi, room_size=max_size, entry_point="minigrid.envs:MultiRoomEnv", kwargs=kw)


_GOTO_OBJ_MISSIONS = tuple(f"go to {a} {c} {t}" for a in ("the", "a") for c in _COLOR_NAMES for t in ("key", "ball", "box"))


def _babyai_goto(id_, kind, room_size, num_dists, missions, cls, kwargs=None):
    # RoomGridLevel (envs/babyai/core/roomgrid_level.py:60-85) on a 1x1 RoomGrid: per-episode max_steps =
    # num_navs(1) * room_size**2; rows minigrid/__init__.py:572-679, 750-753
    return EnvSpec(id_, kind, room_size, room_size, room_size * room_size, False, missions, num_dists=num_dists,
                   entry_point=f"minigrid.envs.babyai:{cls}", kwargs=kwargs or {})


# "pick up " + ObjDesc.surface (envs/babyai/core/verifier.py:73-103): article x (no colour | colour) x ("object" | type)
_PUTNEXT_MISSIONS = tuple(f"put the {c1} {t1} next to the {c2} {t2}" for c1 in _COLOR_NAMES for t1 in ("key", "ball", "box")
                          for c2 in _COLOR_NAMES for t2 in ("key", "ball", "box"))
_PICKUP_MISSIONS = tuple("pick up " + art + " " + (c + " " if c else "") + t for art in ("the", "a")
                         for c in ("",) + _COLOR_NAMES for t in ("object", "key", "ball", "box"))


def _babyai_pickup(id_, kind, room_size, cls, kwargs=None):
    # one-room RoomGridLevel with a PickupInstr: max_steps = room_size**2 (roomgrid_level.py:71-85);
    # rows minigrid/__init__.py:865-873, 1060-1080
    return EnvSpec(id_, kind, room_size, room_size, room_size * room_size, False, _PICKUP_MISSIONS, room_size=room_size,
                   entry_point=f"minigrid.envs.babyai:{cls}", kwargs=kwargs or {})


_ROWS = [
    _empty("MiniGrid-Empty-5x5-v0", 5), _empty("MiniGrid-Empty-Random-5x5-v0", 5, True),
    _empty("MiniGrid-Empty-6x6-v0", 6), _empty("MiniGrid-Empty-Random-6x6-v0", 6, True),
    _empty("MiniGrid-Empty-8x8-v0", 8), _empty("MiniGrid-Empty-16x16-v0", 16),
    _doorkey("MiniGrid-DoorKey-5x5-v0", 5), _doorkey("MiniGrid-DoorKey-6x6-v0", 6),
    _doorkey("MiniGrid-DoorKey-8x8-v0", 8), _doorkey("MiniGrid-DoorKey-16x16-v0", 16),
    _crossing("MiniGrid-LavaCrossingS9N1-v0", 9, 1), _crossing("MiniGrid-LavaCrossingS9N2-v0", 9, 2),
    _crossing("MiniGrid-LavaCrossingS9N3-v0", 9, 3), _crossing("MiniGrid-LavaCrossingS11N5-v0", 11, 5),
    _crossing("MiniGrid-SimpleCrossingS9N1-v0", 9, 1, False), _crossing("MiniGrid-SimpleCrossingS9N2-v0", 9, 2, False),
    _crossing("MiniGrid-SimpleCrossingS9N3-v0", 9, 3, False), _crossing("MiniGrid-SimpleCrossingS11N5-v0", 11, 5, False),
    _goto_red_ball("BabyAI-GoToRedBall-v0", 7), _goto_red_ball("BabyAI-GoToRedBallNoDists-v0", 0),
    _lavagap("MiniGrid-LavaGapS5-v0", 5), _lavagap("MiniGrid-LavaGapS6-v0", 6), _lavagap("MiniGrid-LavaGapS7-v0", 7),
    _distshift("MiniGrid-DistShift1-v0", 2), _distshift("MiniGrid-DistShift2-v0", 5),
    # envs/fourrooms.py:59-73: 19x19, max_steps=100, default see_through_walls=False; row minigrid/__init__.py:213-216
    EnvSpec("MiniGrid-FourRooms-v0", ENV_FOURROOMS, 19, 19, 100, False, ("reach the goal",),
            entry_point="minigrid.envs:FourRoomsEnv"),
    _fetch("MiniGrid-Fetch-5x5-N2-v0", 5, 2), _fetch("MiniGrid-Fetch-6x6-N2-v0", 6, 2), _fetch("MiniGrid-Fetch-8x8-N3-v0", 8, 3),
    # envs/lockedroom.py:82-102 (size 19, max_steps = 10 * size); row minigrid/__init__.py:312-319
    EnvSpec("MiniGrid-LockedRoom-v0", ENV_LOCKEDROOM, 19, 19, 190, False,
            tuple(f"get the {a} key from the {b} room, unlock the {a} door and go to the goal" for a in _COLOR_NAMES for b in _COLOR_NAMES),
            entry_point="minigrid.envs:LockedRoomEnv", kwargs={}),
    # envs/playground.py:16-29 (19x19, max_steps 100, the mission is the empty string); row minigrid/__init__.py:516-523
    EnvSpec("MiniGrid-Playground-v0", ENV_PLAYGROUND, 19, 19, 100, False, ("",), entry_point="minigrid.envs:PlaygroundEnv", kwargs={}),
    _multiroom("MiniGrid-MultiRoom-N2-S4-v0", 2, 2, 4), _multiroom("MiniGrid-MultiRoom-N4-S5-v0", 6, 6, 5),
    _multiroom("MiniGrid-MultiRoom-N4-S5-v1", 4, 4, 5), _multiroom("MiniGrid-MultiRoom-N6-v0", 6, 6),
    _gotoobject("MiniGrid-GoToObject-6x6-N2-v0", 6, 2), _gotoobject("MiniGrid-GoToObject-8x8-N2-v0", 8, 2),
    _gotodoor("MiniGrid-GoToDoor-5x5-v0", 5), _gotodoor("MiniGrid-GoToDoor-6x6-v0", 6), _gotodoor("MiniGrid-GoToDoor-8x8-v0", 8),
    # unlock.py:52-70 (room_size 6, max_steps 8*36), unlockpickup.py:57-80, blockedunlockpickup.py:65-88 (16*36);
    # rows minigrid/__init__.py:17-21,555,560-563
    # redbluedoors.py:60-76 (width 2*size, max_steps 20*size**2); rows minigrid/__init__.py:541-549
    EnvSpec("MiniGrid-RedBlueDoors-6x6-v0", ENV_REDBLUEDOORS, 12, 6, 20 * 36, False, ("open the red door then the blue door",),
            entry_point="minigrid.envs:RedBlueDoorEnv", kwargs={"size": 6}),
    EnvSpec("MiniGrid-RedBlueDoors-8x8-v0", ENV_REDBLUEDOORS, 16, 8, 20 * 64, False, ("open the red door then the blue door",),
            entry_point="minigrid.envs:RedBlueDoorEnv"),
    # memory.py:69-90 (max_steps 5*size**2, see_through_walls=False); rows minigrid/__init__.py:323-357
    *[EnvSpec(f"MiniGrid-MemoryS{sz}{'Random' if rnd else ''}-v0", ENV_MEMORY, sz, sz, 5 * sz * sz, False,
              ("go to the matching object at the end of the hallway",), random_length=rnd,
              entry_point="minigrid.envs:MemoryEnv", kwargs={"size": sz, **({"random_length": True} if rnd else {})})
      for sz, rnd in ((17, True), (13, True), (13, False), (11, False), (9, False), (7, False))],
    # keycorridor.py:75-104 (num_cols 3, max_steps 30*room_size**2, obj_type "ball"); rows minigrid/__init__.py:255-289
    *[EnvSpec(f"MiniGrid-KeyCorridorS{rs}R{rows}-v0", ENV_KEYCORRIDOR, 3 * (rs - 1) + 1, rows * (rs - 1) + 1, 30 * rs * rs, False,
              tuple(f"pick up the {c} ball" for c in _COLOR_NAMES), room_size=rs, entry_point="minigrid.envs:KeyCorridorEnv",
              kwargs={"room_size": rs, "num_rows": rows})
      for rs, rows in ((3, 1), (3, 2), (3, 3), (4, 3), (5, 3), (6, 3))],
    _babyai_pickup("BabyAI-PickupDist-v0", ENV_PICKUPDIST, 7, "PickupDist"),
    _babyai_pickup("BabyAI-PickupDistDebug-v0", ENV_PICKUPDIST_DEBUG, 7, "PickupDist", {"debug": True}),
    _babyai_pickup("BabyAI-OneRoomS8-v0", ENV_ONEROOM, 8, "OneRoomS8"),
    *[_babyai_pickup(f"BabyAI-OneRoomS{s_}-v0", ENV_ONEROOM, s_, "OneRoomS8", {"room_size": s_}) for s_ in (12, 16, 20)],
    # envs/babyai/other.py:163-177: 3 x 3 rooms, max_steps = 20 * room_size**2; rows minigrid/__init__.py:1001-1016
    *[EnvSpec(f"BabyAI-FindObjS{rs}-v0", ENV_FINDOBJ, 3 * (rs - 1) + 1, 3 * (rs - 1) + 1, 20 * rs * rs, False, _PICKUP_MISSIONS,
              room_size=rs, entry_point="minigrid.envs.babyai:FindObjS5", kwargs={} if rs == 5 else {"room_size": rs})
      for rs in (5, 6, 7)],
    # envs/babyai/unlock.py:163-174: RoomGridLevel defaults (3 x 3 rooms of size 8), max_steps = 1 * 64 * 9; rows minigrid/__init__.py:956-965
    EnvSpec("BabyAI-UnlockLocal-v0", ENV_UNLOCKLOCAL, 22, 22, 576, False, ("open the door",), room_size=8, num_dists=0,
            entry_point="minigrid.envs.babyai:UnlockLocal", kwargs={}),
    EnvSpec("BabyAI-UnlockLocalDist-v0", ENV_UNLOCKLOCAL, 22, 22, 576, False, ("open the door",), room_size=8, num_dists=3,
            entry_point="minigrid.envs.babyai:UnlockLocal", kwargs={"distractors": True}),
    # envs/babyai/other.py:231-250: 3 columns x num_rows rooms, max_steps = 30 * room_size**2; rows minigrid/__init__.py:1018-1057
    *[EnvSpec(name, ENV_BABYAI_KEYCORRIDOR, 3 * (rs - 1) + 1, rows * (rs - 1) + 1, 30 * rs * rs, False, _PICKUP_MISSIONS, room_size=rs,
              entry_point="minigrid.envs.babyai:KeyCorridor", kwargs=kw)
      for name, rs, rows, kw in (("BabyAI-KeyCorridor-v0", 6, 3, {}),
                                 *[(f"BabyAI-KeyCorridorS{a}R{b}-v0", a, b, {"room_size": a, "num_rows": b})
                                   for a, b in ((3, 1), (3, 2), (3, 3), (4, 3), (5, 3), (6, 3))])],
    # envs/babyai/open.py:140-146: 1 x 2 rooms of size 5, max_steps = 1 * 25 * 2; row minigrid/__init__.py:773-776
    EnvSpec("BabyAI-OpenRedDoor-v0", ENV_OPENREDDOOR, 9, 5, 50, False, ("open the red door",), room_size=5,
            entry_point="minigrid.envs.babyai:OpenRedDoor", kwargs={}),
    _babyai_goto("BabyAI-GoToRedBallGrey-v0", ENV_GOTO_REDBALLGREY, 8, 7, ("go to the red ball", "go to a red ball"), "GoToRedBallGrey"),
    _babyai_goto("BabyAI-GoToRedBlueBall-v0", ENV_GOTO_REDBLUEBALL, 8, 7, ("go to the red ball", "go to the blue ball"), "GoToRedBlueBall"),
    _babyai_goto("BabyAI-GoToObj-v0", ENV_GOTO_OBJ, 8, 1, _GOTO_OBJ_MISSIONS, "GoToObj"),
    _babyai_goto("BabyAI-GoToObjS4-v0", ENV_GOTO_OBJ, 4, 1, _GOTO_OBJ_MISSIONS, "GoToObj", {"room_size": 4}),
    _babyai_goto("BabyAI-GoToObjS6-v1", ENV_GOTO_OBJ, 6, 1, _GOTO_OBJ_MISSIONS, "GoToObj", {"room_size": 6}),
    _babyai_goto("BabyAI-GoToLocal-v0", ENV_GOTO_LOCAL, 8, 8, _GOTO_OBJ_MISSIONS, "GoToLocal"),
    *[_babyai_goto(f"BabyAI-GoToLocalS{s_}N{n_}-v0", ENV_GOTO_LOCAL, s_, n_, _GOTO_OBJ_MISSIONS, "GoToLocal",
                   {"room_size": s_, "num_dists": n_})
      for s_, n_ in ((5, 2), (6, 2), (6, 3), (6, 4), (7, 4), (7, 5), (8, 2), (8, 3), (8, 4), (8, 5), (8, 6), (8, 7))],
    _dynobs("MiniGrid-Dynamic-Obstacles-5x5-v0", 5, 2), _dynobs("MiniGrid-Dynamic-Obstacles-Random-5x5-v0", 5, 2, True),
    _dynobs("MiniGrid-Dynamic-Obstacles-6x6-v0", 6, 3), _dynobs("MiniGrid-Dynamic-Obstacles-Random-6x6-v0", 6, 3, True),
    _dynobs("MiniGrid-Dynamic-Obstacles-8x8-v0", 8, 4), _dynobs("MiniGrid-Dynamic-Obstacles-16x16-v0", 16, 8),
    # envs/obstructedmaze.py:80-106, obstructedmaze_v1.py: room_size 6, max_steps = 4 * num_rooms_visited * 36; rows
    # minigrid/__init__.py:390-515.  flags (num_crossings) = key_in_box | blocked << 1 | v1 class << 2 | 1 x 2 class << 3
    *[_obstructedmaze(*r) for r in (
        ("1Dl-v0", 1, 2, 2, False, False, 1, (0, 0), False, True), ("1Dlh-v0", 1, 2, 2, True, False, 1, (0, 0), False, True),
        ("1Dlhb-v0", 1, 2, 2, True, True, 1, (0, 0), False, True),
        ("2Dl-v0", 3, 3, 4, False, False, 1, (2, 1), False, False), ("2Dlh-v0", 3, 3, 4, True, False, 1, (2, 1), False, False),
        ("2Dlhb-v0", 3, 3, 4, True, True, 1, (2, 1), False, False), ("1Q-v0", 3, 3, 5, True, True, 1, (1, 1), False, False),
        ("2Q-v0", 3, 3, 11, True, True, 2, (2, 1), False, False), ("Full-v0", 3, 3, 25, True, True, 4, (1, 1), False, False),
        ("2Dlhb-v1", 3, 3, 4, True, True, 1, (2, 1), True, False), ("1Q-v1", 3, 3, 5, True, True, 1, (1, 1), True, False),
        ("2Q-v1", 3, 3, 11, True, True, 2, (2, 1), True, False), ("Full-v1", 3, 3, 25, True, True, 4, (1, 1), True, False))],
    # envs/putnear.py:68-93: see_through_walls=True, max_steps = 5 * size; rows minigrid/__init__.py:526-537.  324 missions in the
    # order of the placeholders (move colour, move type, target colour, target type): 16-bit mission ids
    *[EnvSpec(name, ENV_PUTNEAR, size, size, 5 * size, True,
              tuple(f"put the {mc} {mt} near the {tc} {tt}" for mc in _COLOR_NAMES for mt in ("key", "ball", "box")
                    for tc in _COLOR_NAMES for tt in ("key", "ball", "box")),
              num_dists=n, entry_point="minigrid.envs:PutNearEnv", kwargs={} if size == 6 else {"size": size, "numObjs": n})
      for name, size, n in (("MiniGrid-PutNear-6x6-N2-v0", 6, 2), ("MiniGrid-PutNear-8x8-N3-v0", 8, 3))],
    # the multi-room BabyAI levels with one instruction: goto.py:403-426 (GoTo), pickup.py:66-72, open.py:69-86; max_steps =
    # room_size**2 * rows * cols (roomgrid_level.py:71-85, one instruction); rows minigrid/__init__.py:681-731, 760-763, 848-851
    *[EnvSpec(name, ENV_BABYAI_GOTO, cols * (rs - 1) + 1, rows * (rs - 1) + 1, rs * rs * rows * cols, False, _GOTO_OBJ_MISSIONS,
              num_crossings=int(opened), num_dists=nd, room_size=rs, entry_point="minigrid.envs.babyai:GoTo", kwargs=kw)
      for name, rs, rows, cols, nd, opened, kw in (
          ("BabyAI-GoTo-v0", 8, 3, 3, 18, False, {}), ("BabyAI-GoToOpen-v0", 8, 3, 3, 18, True, {"doors_open": True}),
          ("BabyAI-GoToObjMaze-v0", 8, 3, 3, 1, False, {"num_dists": 1, "doors_open": False}),
          ("BabyAI-GoToObjMazeOpen-v0", 8, 3, 3, 1, True, {"num_dists": 1, "doors_open": True}),
          ("BabyAI-GoToObjMazeS4R2-v0", 4, 2, 2, 1, False, {"num_dists": 1, "room_size": 4, "num_rows": 2, "num_cols": 2}),
          *[(f"BabyAI-GoToObjMazeS{s_}-v0", s_, 3, 3, 1, False, {"num_dists": 1, "room_size": s_}) for s_ in (4, 5, 6, 7)])],
    EnvSpec("BabyAI-Pickup-v0", ENV_BABYAI_PICKUP, 22, 22, 576, False, _PICKUP_MISSIONS, num_dists=18, room_size=8,
            entry_point="minigrid.envs.babyai:Pickup", kwargs={}),
    EnvSpec("BabyAI-Open-v0", ENV_BABYAI_OPEN, 22, 22, 576, False,
            tuple(f"open {art} {c} door" for art in ("the", "a") for c in _COLOR_NAMES), num_dists=18, room_size=8,
            entry_point="minigrid.envs.babyai:Open", kwargs={}),
    # envs/babyai/unlock.py (UnlockPickup(-Dist) :307-319, BlockedUnlockPickup :380-393, UnlockToUnlock :452-474, Unlock :67-112),
    # goto.py (GoToDoor :730-740, GoToObjDoor :800-813, GoToImpUnlock :486-531), pickup.py (UnblockPickup :128-140, PickupAbove
    # :354-362); rows minigrid/__init__.py.  max_steps: the class's own value, or room_size**2 * rooms for one instruction
    *[EnvSpec(name, kind, cols * (rs - 1) + 1, rows * (rs - 1) + 1, ms, False, missions, num_dists=nd, room_size=rs,
              entry_point="minigrid.envs.babyai:" + cls, kwargs=kw)
      for name, kind, cls, rs, rows, cols, ms, nd, missions, kw in (
          ("BabyAI-UnlockPickup-v0", ENV_BABYAI_UNLOCKPICKUP, "UnlockPickup", 6, 1, 2, 72, 0, _PICKUP_MISSIONS, {}),
          ("BabyAI-UnlockPickupDist-v0", ENV_BABYAI_UNLOCKPICKUP, "UnlockPickup", 6, 1, 2, 72, 4, _PICKUP_MISSIONS, {"distractors": True}),
          ("BabyAI-BlockedUnlockPickup-v0", ENV_BABYAI_BLOCKEDUNLOCKPICKUP, "BlockedUnlockPickup", 6, 1, 2, 576, 0, _PICKUP_MISSIONS, {}),
          ("BabyAI-UnlockToUnlock-v0", ENV_UNLOCKTOUNLOCK, "UnlockToUnlock", 6, 1, 3, 1080, 0, _PICKUP_MISSIONS, {}),
          ("BabyAI-KeyInBox-v0", ENV_KEYINBOX, "KeyInBox", 8, 3, 3, 576, 0, ("open the door",), {}),
          ("BabyAI-Unlock-v0", ENV_BABYAI_UNLOCK, "Unlock", 8, 3, 3, 576, 0,
           tuple(f"open {art} {c} door" for art in ("the", "a") for c in _COLOR_NAMES), {}),
          ("BabyAI-GoToDoor-v0", ENV_BABYAI_GOTODOOR, "GoToDoor", 7, 3, 3, 441, 0,
           tuple(f"go to {art} {c} door" for art in ("the", "a") for c in _COLOR_NAMES), {}),
          ("BabyAI-GoToObjDoor-v0", ENV_GOTOOBJDOOR, "GoToObjDoor", 8, 3, 3, 576, 0,
           tuple(f"go to {art} {c} {t}" for art in ("the", "a") for c in _COLOR_NAMES for t in ("key", "ball", "box", "door")), {}),
          ("BabyAI-GoToImpUnlock-v0", ENV_GOTOIMPUNLOCK, "GoToImpUnlock", 8, 3, 3, 576, 0, _GOTO_OBJ_MISSIONS, {}),
          ("BabyAI-UnblockPickup-v0", ENV_UNBLOCKPICKUP, "UnblockPickup", 8, 3, 3, 576, 0, _PICKUP_MISSIONS, {}),
          ("BabyAI-PickupAbove-v0", ENV_PICKUPABOVE, "PickupAbove", 6, 3, 3, 288, 0, _PICKUP_MISSIONS, {}))],
    # envs/babyai/putnext.py:68-80 (PutNextLocal: one room, max_steps = 2 * room_size**2 for its one PutNextInstr), :148-214 (PutNext: 1 x 2
    # rooms, max_steps = 8 * room_size**2); 324 missions = (move colour, move type, fixed colour, fixed type); rows minigrid/__init__.py
    *[EnvSpec(name, ENV_PUTNEXTLOCAL, rs, rs, 2 * rs * rs, False, _PUTNEXT_MISSIONS, num_dists=n, room_size=rs,
              entry_point="minigrid.envs.babyai:PutNextLocal", kwargs=kw)
      for name, rs, n, kw in (("BabyAI-PutNextLocal-v0", 8, 8, {}), ("BabyAI-PutNextLocalS5N3-v0", 5, 3, {"room_size": 5, "num_objs": 3}),
                              ("BabyAI-PutNextLocalS6N4-v0", 6, 4, {"room_size": 6, "num_objs": 4}))],
    *[EnvSpec(f"BabyAI-PutNextS{rs}N{n}{'Carrying' if carrying else ''}-v0", ENV_PUTNEXT, 2 * (rs - 1) + 1, rs, 8 * rs * rs, False, _PUTNEXT_MISSIONS,
              num_dists=n, room_size=rs, num_crossings=int(carrying), entry_point="minigrid.envs.babyai:PutNext",
              kwargs={"room_size": rs, "objs_per_room": n, **({"start_carrying": True} if carrying else {})})
      for rs, n, carrying in ((4, 1, False), (5, 2, False), (5, 1, False), (6, 3, False), (7, 4, False), (5, 2, True), (6, 3, True), (7, 4, True))],
    # envs/babyai/other.py:83-106 (room_size 7), open.py:203-229 (select_by None | "color" | "loc"; debug = strict OpenInstr)
    EnvSpec("BabyAI-ActionObjDoor-v0", ENV_ACTIONOBJDOOR, 19, 19, 441, False,
            tuple(f"{verb} {art} {c} {t}" for verb in ("go to", "pick up", "open") for art in ("the", "a") for c in _COLOR_NAMES
                  for t in ("key", "ball", "box", "door")), room_size=7, entry_point="minigrid.envs.babyai:ActionObjDoor", kwargs={}),
    *[EnvSpec(name, ENV_OPENDOOR, 22, 22, 576, False,
              tuple(f"open the {c} door" for c in _COLOR_NAMES) +
              tuple(f"open {art} door {loc}" for art in ("the", "a") for loc in ("on your left", "on your right", "in front of you", "behind you")),
              room_size=8, num_crossings=sel, strip2_row=int(dbg), entry_point="minigrid.envs.babyai:OpenDoor", kwargs=kw)
      for name, sel, dbg, kw in (("BabyAI-OpenDoor-v0", 0, False, {}), ("BabyAI-OpenDoorDebug-v0", 0, True, {"debug": True, "select_by": None}),
                                 ("BabyAI-OpenDoorColor-v0", 1, False, {"select_by": "color"}), ("BabyAI-OpenDoorLoc-v0", 2, False, {"select_by": "loc"}))],
    # the sentence levels: the mission is an instruction tree, delivered as data and turned into the sentence on the host
    # (minigrid_amd/sentence.py); no mission-id table.  envs/babyai/open.py:289-325 (OpenTwoDoors; first / second colour as COLOR_NAMES
    # indices in agent_start), :383-425 (OpenDoorsOrder), other.py:388-428 (MoveTwoAcross): fixed max_steps
    *[EnvSpec(name, ENV_OPENTWODOORS, 16, 16, 720, False, ("",), room_size=6, agent_start=(c1, c2, 0), strip2_row=int(strict),
              entry_point="minigrid.envs.babyai:OpenTwoDoors", kwargs=kw)
      for name, c1, c2, strict, kw in (("BabyAI-OpenTwoDoors-v0", -1, -1, False, {}),
                                       ("BabyAI-OpenRedBlueDoors-v0", 4, 0, False, {"first_color": "red", "second_color": "blue"}),
                                       ("BabyAI-OpenRedBlueDoorsDebug-v0", 4, 0, True, {"first_color": "red", "second_color": "blue", "strict": True}))],
    *[EnvSpec(f"BabyAI-OpenDoorsOrderN{n}{'Debug' if dbg else ''}-v0", ENV_OPENDOORSORDER, 16, 16, 720, False, ("",), room_size=6, num_dists=n,
              strip2_row=int(dbg), entry_point="minigrid.envs.babyai:OpenDoorsOrder", kwargs={"num_doors": n, **({"debug": True} if dbg else {})})
      for n, dbg in ((2, False), (4, False), (2, True), (4, True))],
    *[EnvSpec(f"BabyAI-MoveTwoAcrossS{rs}N{n}-v0", ENV_MOVETWOACROSS, 2 * (rs - 1) + 1, rs, 16 * rs * rs, False, ("",), room_size=rs, num_dists=n,
              entry_point="minigrid.envs.babyai:MoveTwoAcross", kwargs={"room_size": rs, "objs_per_room": n})
      for rs, n in ((5, 2), (8, 9))],
    # LevelGen (envs/babyai/core/levelgen.py:24-80) configurations: pickup.py:198-213, goto.py:590-606, synth.py:83-97, :168-178, :274-281,
    # :374-382, :476-480, :570-576.  max_steps is per episode (num_navs * room_size**2 * rooms, applied on the device); the value here is
    # what the reference's env holds after gym.make + reset(seed=0) (tests/golden/reference_registry.json).  BabyAI-SynthS5R2-v0: the
    # reference never returns from about 0.4 % of its resets (RoomGrid.place_agent, roomgrid.py:327-332); the device decides that case
    # exactly and raises RecursionError when an env reaches such an episode (`stuck_place_agent="redraw"` accepts a redrawn map)
    *[EnvSpec(name, ENV_LEVELGEN, cols * (rs - 1) + 1, rows * (rs - 1) + 1, ms, False, ("",), room_size=rs, num_dists=nd, strip2_row=pct,
              num_crossings=acts | kinds << 4 | int(loc) << 7 | int(unb) << 8 | int(imp) << 9,
              entry_point="minigrid.envs.babyai:" + cls, kwargs=kw)
      for name, cls, rs, rows, cols, nd, acts, kinds, loc, unb, imp, pct, ms, kw in (
          ("BabyAI-PickupLoc-v0", "PickupLoc", 8, 1, 1, 8, 0b0010, 0b001, True, False, True, 0, 64, {}),
          ("BabyAI-GoToSeq-v0", "GoToSeq", 8, 3, 3, 18, 0b0001, 0b111, False, False, True, 0, 2304, {}),
          ("BabyAI-GoToSeqS5R2-v0", "GoToSeq", 5, 2, 2, 4, 0b0001, 0b111, False, False, True, 0, 100,
           {"room_size": 5, "num_rows": 2, "num_cols": 2, "num_dists": 4}),
          ("BabyAI-Synth-v0", "Synth", 8, 3, 3, 18, 0b1111, 0b001, False, True, False, 50, 1152, {}),
          ("BabyAI-SynthS5R2-v0", "Synth", 5, 2, 3, 18, 0b1111, 0b001, False, True, False, 50, 150, {"room_size": 5, "num_rows": 2}),
          ("BabyAI-SynthLoc-v0", "SynthLoc", 8, 3, 3, 18, 0b1111, 0b001, True, True, False, 50, 1152, {}),
          ("BabyAI-SynthSeq-v0", "SynthSeq", 8, 3, 3, 18, 0b1111, 0b111, True, True, False, 50, 2880, {}),
          ("BabyAI-MiniBossLevel-v0", "MiniBossLevel", 5, 2, 2, 7, 0b1111, 0b111, True, True, True, 25, 100, {}),
          ("BabyAI-BossLevel-v0", "BossLevel", 8, 3, 3, 18, 0b1111, 0b111, True, True, True, 50, 2880, {}),
          ("BabyAI-BossLevelNoUnlock-v0", "BossLevelNoUnlock", 8, 3, 3, 18, 0b1111, 0b111, True, True, False, 0, 2880, {}))],
    _roomgrid_1x2("MiniGrid-Unlock-v0", ENV_UNLOCK, 6, 8 * 36, ("open the door",), "minigrid.envs:UnlockEnv"),
    _roomgrid_1x2("MiniGrid-UnlockPickup-v0", ENV_UNLOCKPICKUP, 6, 8 * 36,
                  tuple(f"pick up the {c} box" for c in _COLOR_NAMES), "minigrid.envs:UnlockPickupEnv"),
    _roomgrid_1x2("MiniGrid-BlockedUnlockPickup-v0", ENV_BLOCKEDUNLOCKPICKUP, 6, 16 * 36,
                  tuple(f"pick up the {c} {t}" for c in _COLOR_NAMES for t in ("box", "key")),
                  "minigrid.envs:BlockedUnlockPickupEnv"),
]

# registrations that rely on the classes' defaults (minigrid/__init__.py:36-39, 226-229, 240-244, 131-134, 584-587): the rows
# above spell the defaults out; the registered kwargs / entry points are these
_AS_REGISTERED = {
    "MiniGrid-Empty-8x8-v0": ("minigrid.envs:EmptyEnv", {}),
    "MiniGrid-Fetch-8x8-N3-v0": ("minigrid.envs:FetchEnv", {}),
    "MiniGrid-GoToObject-6x6-N2-v0": ("minigrid.envs:GoToObjectEnv", {}),
    "MiniGrid-GoToDoor-5x5-v0": ("minigrid.envs:GoToDoorEnv", {}),
    "MiniGrid-Dynamic-Obstacles-8x8-v0": ("minigrid.envs:DynamicObstaclesEnv", {}),
    "BabyAI-GoToRedBallNoDists-v0": ("minigrid.envs.babyai:GoToRedBallNoDists", {}),
}
_ROWS = [replace(r, entry_point=_AS_REGISTERED[r.id][0], kwargs=_AS_REGISTERED[r.id][1]) if r.id in _AS_REGISTERED else r for r in _ROWS]

registry: Dict[str, EnvSpec] = {r.id: r for r in _ROWS}


def spec(env_id: str) -> EnvSpec:
    try:
        return registry[env_id]
    except KeyError:
        raise KeyError(
            f"{env_id!r} is not on the accelerated path. Supported ids: {sorted(registry)}") from None
