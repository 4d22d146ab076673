/*
 * minigrid_hip.h — C ABI of libminigrid_hip.so, the MI355X-native lockstep-batched MiniGrid hot path.
 *
 * The reference (Farama-Foundation/Minigrid, pure Python) has no FFI of its own: its extension boundary is the
 * Gymnasium Env protocol.  This header is the boundary a binding for that protocol calls into; each entry point
 * cites the reference interface it replaces (paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add; minigrid_amd/_binding.py is that stub in this repo.
 *
 * Conventions
 *   - plain C types only; every function returns MG_OK (0) or a negative mg_status; mg_last_error() has the text.
 *   - one mg_env = one device + one HIP stream + N lockstep environments.  Calls on one handle are not
 *     re-entrant; different handles may be driven from different host threads.
 *   - the library owns all device buffers.  Output device pointers (mg_get_outputs) are borrowed, stay valid for
 *     the life of the handle, and their CONTENT is overwritten by the next mg_step / mg_reset / mg_rollout.
 *   - mg_step / mg_reset / mg_rollout are asynchronous with respect to the host; mg_copy_outputs and mg_sync
 *     synchronise with the handle's stream.
 *   - all "image" tensors use the reference's index order image[x][y][channel] (core/grid.py:252-266).
 */
#ifndef MINIGRID_HIP_H
#define MINIGRID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 3

#if defined(__GNUC__)
#define MG_API __attribute__((visibility("default")))
#else
#define MG_API
#endif

typedef enum mg_status {
  MG_OK = 0,
  MG_ERR_INVALID = -1,     /* bad argument / unsupported configuration (reference: assert / TypeError)      */
  MG_ERR_HIP = -2,         /* a HIP runtime call failed                                                       */
  MG_ERR_BAD_ACTION = -3,  /* an action outside 0..6 was seen (reference: ValueError, minigrid_env.py:584-585) */
  MG_ERR_GENERATOR = -4,   /* map generation exhausted its retry bound (reference: RecursionError,
                              minigrid_env.py:342-343; roomgrid_level.py:131-134 retries instead), or the episode an env
                              has just started is one whose drawing met RoomGrid.place_agent's unbounded loop
                              (roomgrid.py:327-332: every free cell of the agent's room faces an object; the reference
                              never returns from that reset() -- BabyAI-SynthS5R2-v0, about 0.4 % of the episodes).
                              Episodes are drawn ahead, so the error is held back until the env takes that episode.
                              LevelGen's redraw_stuck bit accepts the redrawn map instead (not a reference behaviour) */
  MG_ERR_NO_DEVICE = -5,   /* no usable HIP device                                                            */
  MG_ERR_OOB = -6,         /* front cell outside the grid (reference: AssertionError, core/grid.py:74-78)     */
  MG_ERR_TRACKED = -7      /* multi-room GoTo: more than four described objects were removed from the grid between two drop actions
                              (the device keeps four stale tracked positions; reported, never silently dropped)              */
} mg_status;

/* Map generators = the reference's `_gen_grid` implementations on the path. */
typedef enum mg_env_kind {
  MG_ENV_EMPTY = 0,         /* envs/empty.py:97-114                                           */
  MG_ENV_DOORKEY = 1,       /* envs/doorkey.py:74-99                                          */
  MG_ENV_CROSSING = 2,      /* envs/crossing.py:131-188 (LavaCrossing / SimpleCrossing)       */
  MG_ENV_GOTO_REDBALL = 3,  /* envs/babyai/goto.py:133-141 + core/roomgrid.py + roomgrid_level.py:119-144 */
  MG_ENV_LAVAGAP = 4,       /* envs/lavagap.py:100-135                                        */
  MG_ENV_DISTSHIFT = 5,     /* envs/distshift.py:103-124                                      */
  MG_ENV_FOURROOMS = 6,     /* envs/fourrooms.py:77-130 (agent_pos = goal_pos = None)         */
  MG_ENV_FETCH = 7,         /* envs/fetch.py:107-175 (num_dists = numObjs); mission id = syntax*12 + colour*2 + type  */
  MG_ENV_GOTODOOR = 8,      /* envs/gotodoor.py:92-149; mission id = COLOR_NAMES index of the target door          */
  /* RoomGrid levels (core/roomgrid.py:72-334), width = (room_size-1)*num_cols+1, height likewise: */
  MG_ENV_UNLOCK = 9,                /* envs/unlock.py:75-98                                                        */
  MG_ENV_UNLOCKPICKUP = 10,         /* envs/unlockpickup.py:82-107; mission id = COLOR_NAMES index of the box      */
  MG_ENV_BLOCKEDUNLOCKPICKUP = 11,  /* envs/blockedunlockpickup.py:90-119; mission id = colour index * 2 (box)     */
  MG_ENV_REDBLUEDOORS = 12, /* envs/redbluedoors.py:78-126 (width = 2 * height)                                    */
  MG_ENV_MEMORY = 13,       /* envs/memory.py:92-164 (odd size; random_length)                                     */
  MG_ENV_KEYCORRIDOR = 14,  /* envs/keycorridor.py:106-145 (3 x num_rows RoomGrid, connect_all); mission id = ball colour */
  MG_ENV_GOTO_REDBALLGREY = 16, MG_ENV_GOTO_REDBLUEBALL = 17, MG_ENV_GOTO_OBJ = 18, MG_ENV_GOTO_LOCAL = 19,
                            /* envs/babyai/goto.py:67-78, 661-677, 256-260, 333-338: single-room GoToInstr levels
                               (room_size = width = height in 4..8, num_dists <= 8); GoToObj / GoToLocal mission id =
                               ("a" ? 18 : 0) + COLOR_NAMES index * 3 + (key 0, ball 1, box 2)                     */
  MG_ENV_GOTOOBJECT = 20,   /* envs/gotoobject.py:93-153 (size 4..8, num_dists = numObjs 1..8); mission id = COLOR_NAMES index * 3 + type */
  MG_ENV_LOCKEDROOM = 21,   /* envs/lockedroom.py:104-176 (19 x 19); mission id = locked room's colour index * 6 + key room's            */
  MG_ENV_PLAYGROUND = 22,   /* envs/playground.py:31-91 (19 x 19; no goal: episodes only end by truncation)                              */
  MG_ENV_MULTIROOM = 23,    /* envs/multiroom.py:118-300 (25 x 25): num_crossings = minNumRooms, num_dists = maxNumRooms <= 6,
                               room_size = maxRoomSize                                                                               */
  MG_ENV_PICKUPDIST = 24,   /* envs/babyai/pickup.py:215-290 (one 7x7 room, 5 distractors, PickupInstr); mission id = article * 28 +
                               (no colour 0 | COLOR_NAMES index + 1) * 4 + ("object" 0 | key 1 | ball 2 | box 3)                 */
  MG_ENV_ONEROOM = 25,      /* envs/babyai/other.py:275-332 (OneRoomS8/S12/S16/S20: width = height = room_size)                   */
  MG_ENV_OPENREDDOOR = 26,  /* envs/babyai/open.py:89-146 (1 x 2 rooms, room_size 5, OpenInstr)                                   */
  MG_ENV_PICKUPDIST_DEBUG = 27, /* PickupDist(debug=True): strict PickupInstr, a wrong pickup ends the episode (verifier.py:356-359) */
  MG_ENV_FINDOBJ = 28,      /* envs/babyai/other.py:109-177 (FindObjS5/S6/S7: 3 x 3 rooms, connect_all, PickupInstr by type)             */
  MG_ENV_UNLOCKLOCAL = 29,  /* envs/babyai/unlock.py:114-174 (3 x 3 rooms of room_size 8; num_dists = 0 | 3 distractors = UnlockLocalDist)  */
  MG_ENV_BABYAI_KEYCORRIDOR = 30, /* envs/babyai/other.py:180-272: MG_ENV_KEYCORRIDOR's map with PickupInstr(ObjDesc("ball"))           */
  MG_ENV_OBSTRUCTEDMAZE = 31, /* envs/obstructedmaze.py:111-270, obstructedmaze_v1.py:37-100 (room_size 6; 1 x 2 or 3 x 3 rooms): num_crossings =
                               flags (1 key_in_box | 2 blocked | 4 the v1 class | 8 ObstructedMaze_1Dlhb), num_dists = num_quarters,
                               agent_start_x / agent_start_y = agent_room                                                             */
  MG_ENV_BABYAI_GOTO = 33, MG_ENV_BABYAI_PICKUP = 34, MG_ENV_BABYAI_OPEN = 35,
                            /* envs/babyai/goto.py:403-426 (GoTo, GoToOpen, GoToObjMaze*: num_crossings = doors_open), pickup.py:66-72,
                               open.py:69-86: num_cols x num_rows rooms (2 x 2 / 3 x 3) of room_size 4..8, num_dists distractors over all
                               rooms; mission ids as GoToObj / PickupDist / article * 6 + colour                                  */
  MG_ENV_BABYAI_UNLOCKPICKUP = 36,        /* envs/babyai/unlock.py:307-319 (1 x 2 rooms; num_dists = 0 | 4 = UnlockPickupDist)               */
  MG_ENV_BABYAI_BLOCKEDUNLOCKPICKUP = 37, /* unlock.py:380-393 (1 x 2 rooms)                                                             */
  MG_ENV_UNLOCKTOUNLOCK = 38,             /* unlock.py:452-474 (1 x 3 rooms)                                                             */
  MG_ENV_KEYINBOX = 39,                   /* unlock.py:232-242 (3 x 3 rooms; the locked door's key lies in a box of a random colour)     */
  MG_ENV_BABYAI_UNLOCK = 40,              /* unlock.py:67-112 (3 x 3 rooms, OpenInstr about a door colour; id = article * 6 + colour)    */
  MG_ENV_BABYAI_GOTODOOR = 41,            /* goto.py:730-740 (GoToInstr about a door colour; id = article * 6 + colour)                  */
  MG_ENV_GOTOOBJDOOR = 42,                /* goto.py:800-813 (id = article * 24 + colour * 4 + (key, ball, box, door))                   */
  MG_ENV_UNBLOCKPICKUP = 43,              /* pickup.py:128-140 (20 distractors, rejected while every object is reachable)                */
  MG_ENV_PICKUPABOVE = 44,                /* pickup.py:354-362                                                                           */
  MG_ENV_GOTOIMPUNLOCK = 45,              /* goto.py:486-531 (ids as GoToObj).  36..45: RoomGrids of 1..3 x 1..3 rooms of room_size 4..8 */
  MG_ENV_PUTNEXTLOCAL = 46,               /* envs/babyai/putnext.py:72-80 (one room, num_dists objects; 324 mission ids)                 */
  MG_ENV_PUTNEXT = 47,                    /* putnext.py:168-214 (1 x 2 rooms, num_dists objects per room; num_crossings = start_carrying) */
  MG_ENV_ACTIONOBJDOOR = 48,              /* other.py:86-106 (id = verb * 48 + article * 24 + colour * 4 + (key, ball, box, door))        */
  MG_ENV_OPENDOOR = 49,                   /* open.py:209-229 (num_crossings = select_by: 0 random | 1 colour | 2 location; strip2_row = strict;
                                             id = colour, or 6 + article * 4 + (left, right, front, behind))                            */
  MG_ENV_OPENTWODOORS = 50,               /* envs/babyai/open.py:306-325 (agent_start_x / agent_start_y = first / second door colour as a COLOR_NAMES
                                             index or -1 = drawn; strip2_row = strict): "open the X door, then open the Y door"           */
  MG_ENV_OPENDOORSORDER = 51,             /* open.py:399-425 (num_dists = num_doors 2..4, strip2_row = debug)                            */
  MG_ENV_MOVETWOACROSS = 52,              /* other.py:404-428 (1 x 2 rooms, num_dists = objs_per_room)                                   */
  MG_ENV_LEVELGEN = 53,                   /* envs/babyai/core/levelgen.py:24-211 (PickupLoc, GoToSeq, Synth*, MiniBossLevel, BossLevel*):
                                             num_crossings = action kinds (bit 0 goto, 1 pickup, 2 open, 3 putnext) | instr kinds (bit 4
                                             action, 5 and, 6 seq) | bit 7 locations | bit 8 unblocking | bit 9 implicit_unlock | bit 10
                                             redraw_stuck (see MG_ERR_GENERATOR);
                                             strip2_row = locked_room_prob in percent; num_dists distractors.  50..53 are the "sentence
                                             levels": the mission is an instruction tree (mg_outputs.sentence), max_steps is per episode */
  MG_ENV_PUTNEAR = 32,      /* envs/putnear.py:101-199 (size 5..8, num_dists = numObjs 2..8); mission id (324 of them, hence 16-bit ids) =
                               ((move colour * 3 + move type) * 6 + target colour) * 3 + target type                                  */
  MG_ENV_DYNOBS = 15        /* envs/dynamicobstacles.py:110-167 (num_dists = n_obstacles <= 8, grid <= 16x16); step() moves
                               the obstacles on the env's own stream, so resets are drawn just in time, not ahead      */
} mg_env_kind;

typedef enum mg_obs_mode {
  MG_OBS_PARTIAL = 0,  /* MiniGridEnv.gen_obs (minigrid_env.py:634-650): (N,V,V,3) u8; also ImgObsWrapper (wrappers.py:187-214);
                          V = agent_view_size (7, or any odd 3..15 = ViewSizeWrapper, wrappers.py:629-673)                     */
  MG_OBS_FULL = 1,     /* FullyObsWrapper.observation (wrappers.py:419-426): (N,W,H,3) u8                                       */
  MG_OBS_ONEHOT = 2,   /* OneHotPartialObsWrapper.observation (wrappers.py:267-284): (N,V,V,20) u8                              */
  MG_OBS_SYMBOLIC = 3, /* SymbolicObsWrapper.observation (wrappers.py:763-782): (N,W,H,3) i8 = (x, y, type or -1), agent = 10   */
  MG_OBS_RGB_PARTIAL = 4, /* RGBImgPartialObsWrapper.observation (wrappers.py:376-381) = get_frame(agent_pov=True)
                             (minigrid_env.py:652-666): (N, V*tile_size, V*tile_size, 3) u8                              */
  MG_OBS_RGB = 5       /* RGBImgObsWrapper.observation (wrappers.py:325-331) = get_full_render (minigrid_env.py:668-714):
                          (N, H*tile_size, W*tile_size, 3) u8, the agent's view highlighted when rgb_highlight != 0       */
} mg_obs_mode;

typedef enum mg_autoreset_mode {
  MG_AUTORESET_NEXT_STEP = 0, /* Gymnasium >= 1.0 default: the step after a done resets, ignores its action, reward 0 */
  MG_AUTORESET_DISABLED = 1,  /* never reset implicitly; the caller uses mg_reset with a mask                         */
  MG_AUTORESET_SAME_STEP = 2  /* gymnasium 0.28 / 0.29 vector semantics (the reference pins gymnasium >= 0.28.1): the step that ends an
                               * episode also resets the env -- obs = the new episode's first, reward / flags = the ended one's.
                               * (The ended episode's last observation is not produced in this mode: a caller that wants it -- Gymnasium
                               * 1.x's info["final_obs"] -- steps with NEXT_STEP and resets the finished envs with a masked mg_reset
                               * before the next step; minigrid_amd/vector_env.py does that for final_obs=True.) */
} mg_autoreset_mode;

typedef enum mg_rng_mode {
  MG_RNG_PCG64 = 0,   /* bit-exact numpy Generator(PCG64(SeedSequence(seed))) stream: reset(seed=s) reproduces the
                         reference's layouts for the same seed, across autoresets (minigrid_env.py:125,247-311)  */
  MG_RNG_PHILOX = 1   /* Philox4x32-10 keyed by (seed, episode): same generator algorithm and distribution,
                         different layouts; no carried 128-bit state                                            */
} mg_rng_mode;

typedef enum mg_action_dtype { MG_ACT_U8 = 0, MG_ACT_I32 = 1, MG_ACT_I64 = 2 } mg_action_dtype;

/* Static per-env-id configuration = one row of the reference registry (minigrid/__init__.py) plus the
 * constructor defaults of that env class.  minigrid_amd/registry.py holds the rows. */
typedef struct mg_config {
  int32_t abi_version;        /* MG_ABI_VERSION */
  int32_t env_kind;           /* mg_env_kind */
  int32_t width, height;      /* grid size in cells (minigrid_env.py:99-100) */
  int32_t max_steps;          /* minigrid_env.py:105; BabyAI: roomgrid_level.py:77-83 */
  int32_t see_through_walls;  /* minigrid_env.py:107 */
  int32_t agent_view_size;    /* odd, 3..15; 7 = the reference default (minigrid_env.py:44,66-68), else ViewSizeWrapper */
  int32_t obs_mode;           /* mg_obs_mode */
  int32_t autoreset_mode;     /* mg_autoreset_mode */
  int32_t rng_mode;           /* mg_rng_mode */
  int32_t num_envs;           /* N lockstep envs on this device */
  int32_t agent_start_x, agent_start_y, agent_start_dir; /* Empty / DistShift / DynamicObstacles: fixed start (empty.py:71-72); x < 0 => place_agent() */
  int32_t num_crossings;      /* Crossing (crossing.py:92) */
  int32_t obstacle_type;      /* Crossing / LavaGap: 9 = lava, 2 = wall (crossing.py:93, lavagap.py:69) */
  int32_t num_dists;          /* GoToRedBall num_dists (goto.py:129); Fetch numObjs (fetch.py:67); DynamicObstacles n_obstacles */
  int32_t null_stream_sync;   /* library-created stream only: 1 = blocking stream (hipStreamDefault), i.e. ordered
                                 with the legacy NULL stream a framework such as PyTorch launches on; 0 = non-blocking */
  int32_t strip2_row;         /* DistShift (distshift.py:72) */
  int32_t no_death_mask;      /* NoDeath wrapper (wrappers.py:845-882): bit t = cells of OBJECT_TO_IDX type t do not kill */
  double death_cost;          /* ... and add this to the reward instead (wrappers.py:879-880)                           */
  int32_t room_size;          /* RoomGrid levels (core/roomgrid.py:75)                                                  */
  int32_t random_length;      /* Memory (memory.py:70)                                                                  */
  int64_t env_index_base;     /* global index of env 0 of this shard (multi-GPU: seed = base_seed + global index) */
  int32_t tile_size;          /* RGB modes: pixels per cell, 1..64 (wrappers.py:305, 355: default 8; 4 | 8 | 12 | 16 take the fast blit); else ignored */
  int32_t rgb_highlight;      /* MG_OBS_RGB: MiniGridEnv.highlight (minigrid_env.py:47, 109; default 1)                    */
  int32_t spare_ring;         /* pre-generated episodes kept per env (power of two, 4..256); 0 = default: 256 for the levels whose refill runs
                                 one lane per episode and for the big-grid maze levels at 32 768 envs or more, 128 for the others, halved while the ring would exceed
                                 min(32 GB, a quarter of the free device memory) */
  int32_t traj_slots;         /* trajectory ring slots S (see mg_outputs); 0 = default (32, fewer when a slot is large);
                                 < 0 = -traj_slots preferred, halved like the default while the ring would exceed 2 GB       */
  int32_t babyai_done_actions; /* envs/babyai/core/verifier.py:26 use_done_actions (the reference reads BABYAI_DONE_ACTIONS when it is imported):
                                 only the `done` action reports -- success iff the previous action completed the instruction, failure
                                 otherwise (verifier.py:228-242).  RoomGridLevel-based levels only; ignored elsewhere.  Default 0.
                                 1 = the reference stepped with INTEGER actions (env.step(6), a numpy vector of actions: what
                                 gymnasium.vector.SyncVectorEnv passes); 2 = stepped with the enum member (env.step(env.actions.done)):
                                 AndInstr.verify's `action is self.env.actions.done` branch (verifier.py:561-563) is then taken --
                                 `done` while both halves report failure fails the instruction.                                     */
} mg_config;

/* Borrowed device pointers to the outputs of the last step/reset = slot 0 of the trajectory ring.  The ring has
 * traj_slots slots of slot_bytes each; slot k holds the outputs of the step k calls before the last one (as far as
 * the last mg_rollout / mg_step_many call reaches), at every pointer below + k * slot_bytes.  The first record_bytes of a
 * slot, starting at `obs`, are one contiguous record {obs (N, ...) | scalars (N) x mg_step_scalars [| sentence (N, 2)]}: a multi-GPU
 * consumer moves a whole step with ONE all-gather.  Since ABI 3 the per-env scalars of a step are ONE 16-byte entry per env
 * (mg_step_scalars: the step kernel writes them with one 16-byte store per env instead of six partial-line stores, which cost
 * a tenth of the fused step's time: profiles/r4/attribution_split.txt); the pointers below address the first env's field and advance by
 * scalar_stride (= sizeof(mg_step_scalars)) bytes per env -- strided views, e.g. reward[i] = *(double*)((char*)reward + i * scalar_stride). */
typedef struct mg_step_scalars {
  double reward;        /* 0 or 1 - 0.9*(step_count/max_steps), bit-exact (minigrid_env.py:240-245)      */
  uint8_t terminated;   /* 0/1                                                                           */
  uint8_t truncated;    /* 0/1 (minigrid_env.py:587-588)                                                 */
  uint8_t direction;    /* agent_dir 0..3 (obs["direction"], minigrid_env.py:648)                        */
  uint8_t action;       /* the action the step applied (device-policy rollouts record it here)           */
  uint16_t mission_id;  /* index into the config's mission-string table (obs["mission"])                 */
  uint16_t reserved;    /* 0                                                                             */
} mg_step_scalars;
typedef struct mg_outputs {
  uint8_t* obs;         /* (N, V,V,3) | (N, W,H,3) | (N, V,V,20) u8, (N, W,H,3) i8 or an RGB frame, C-contiguous */
  double* reward;       /* the fields of env 0's mg_step_scalars; env i's at + i * scalar_stride bytes */
  uint8_t* terminated;
  uint8_t* truncated;
  uint8_t* direction;
  uint16_t* mission_id;
  int64_t obs_bytes_per_env;
  int64_t num_envs;
  uint8_t* action;      /* (strided like reward)                                                        */
  int64_t traj_slots;
  int64_t slot_bytes;
  int64_t record_bytes;
  int64_t max_fused_steps; /* steps one k_step launch of mg_rollout(fused) / mg_step_many runs */
  uint64_t* sentence;   /* sentence levels (MG_ENV_OPENTWODOORS .. MG_ENV_LEVELGEN): (N, 2) u64, the mission as data -- word 0: bits [20k, 20k+20)
                           = leaf k (k < 3), [60:63) root; word 1: [0:20) leaf 3, [20:44) three nodes of 8 bits = kind (1 ", then " 2 " after
                           you " 3 " and ") | a << 2 | b << 5, children 0..3 = leaves, 4..6 = nodes.  leaf = verb (go to, pick up, open, put) |
                           desc << 2 | fixed desc << 11 (put X next to Y); desc = type (door key ball box) | colour << 2 (0 none, COLOR_TO_IDX
                           + 1) | loc << 5 (0 none, left right front behind) | article << 8 (1 = "a").  NULL for every other level.        */
  int64_t scalar_stride; /* bytes between consecutive envs' scalars (16 = sizeof(mg_step_scalars))       */
} mg_outputs;

typedef struct mg_env mg_env;

/* gym.make(id) -> Env.__init__ (minigrid_env.py:34-117).  device < 0 => current device.
 * stream: a hipStream_t to run on (borrowed), or NULL to let the library create its own non-blocking stream. */
MG_API int mg_create(const mg_config* cfg, int device, void* stream, mg_env** out);
MG_API int mg_destroy(mg_env* env);                                   /* Env.close() */
/* Change the OBSERVATION part of a live handle's configuration -- obs_mode, agent_view_size, tile_size, rgb_highlight,
 * no_death_mask / death_cost, traj_slots; every other field of `cfg` must equal what mg_create was given -- without touching the
 * environments: grids, agent records, missions, instruction trees, hidden box contents and every env's generator position stay as
 * they are, only the output buffers are re-made (mg_get_outputs must be called again; their contents start zeroed).  This is what
 * composing the reference's observation wrappers does to ONE env object (minigrid/wrappers.py:187-214, 217-426, 629-882: they
 * wrap the same env, mid-episode or not).  Synchronises. */
MG_API int mg_set_obs_config(mg_env* env, const mg_config* cfg);

/* Env.reset(seed=...) for the selected envs (minigrid_env.py:119-157).
 *   seeds: host array [N] of per-env seeds (env i is seeded exactly like `reset(seed=seeds[i])`), or NULL to
 *          continue each env's own stream like `reset()`;  mask: host array [N] u8 (non-zero = reset), or NULL = all.
 * Produces the reset observation for every env in the output buffers (reward 0, flags 0). */
MG_API int mg_reset(mg_env* env, const uint64_t* seeds, const uint8_t* mask);

/* Env.step(action) for all N envs in lockstep (minigrid_env.py:525-595; BabyAI roomgrid_level.py:87-104).
 * actions: N values of `dtype`, on the host (on_device = 0; copied asynchronously) or already on this device. */
MG_API int mg_step(mg_env* env, const void* actions, int dtype, int on_device);

/* T steps under a uniform-random policy generated on the device (Philox4x32-10 keyed by action_seed, global env
 * index and step number) — the benchmark loop of minigrid/benchmark.py:36-43 with random actions.
 * fused = 0: one k_step launch per step, every step writing slot 0 exactly as mg_step does.
 * fused = 1: up to max_fused_steps steps per launch with the grids resident in LDS; step j of the call writes its full
 * outputs (and the action it applied) to trajectory slot (T-1-j) mod traj_slots, so the last step is in slot 0. */
MG_API int mg_rollout(mg_env* env, int T, uint64_t action_seed, int fused);
/* The same fused loop for actions the caller supplies: u8 [T][N], host or device.  Identical results to T mg_step calls. */
MG_API int mg_step_many(mg_env* env, const uint8_t* actions, int T, int on_device);
/* ONE fused launch of T <= max_fused_steps steps of the device policy (mg_rollout's, same action counter) whose step j lands in
 * trajectory slot slot0 - j: the T step records are the contiguous byte range [slot0 - T + 1, slot0] x slot_bytes.  The unit a
 * multi-GPU consumer gathers with one collective per launch while the next launch runs (minigrid_amd/sharded.py rollout_gather);
 * the reference's counterpart is the body of the benchmark loop, minigrid/benchmark.py:36-43. */
MG_API int mg_rollout_block(mg_env* env, int T, uint64_t action_seed, int slot0);

MG_API int mg_get_outputs(mg_env* env, mg_outputs* out);
/* Synchronises the stream, then copies whichever destinations are non-NULL to host memory.
 * Also surfaces device-side error flags (MG_ERR_BAD_ACTION / MG_ERR_GENERATOR / MG_ERR_OOB). */
MG_API int mg_copy_outputs(mg_env* env, uint8_t* obs, double* reward, uint8_t* terminated, uint8_t* truncated,
                    uint8_t* direction, uint16_t* mission_id);
/* mg_copy_outputs for trajectory slot `slot` (0 = the last step), plus the recorded actions. */
/* sentence levels: the (N, 2) u64 mission words of trajectory slot `slot` (see mg_outputs.sentence) to the host */
MG_API int mg_copy_sentence(mg_env* env, int slot, uint64_t* out);
MG_API int mg_copy_slot(mg_env* env, int slot, uint8_t* obs, double* reward, uint8_t* terminated, uint8_t* truncated,
                 uint8_t* direction, uint16_t* mission_id, uint8_t* action);
MG_API int mg_sync(mg_env* env);        /* synchronises the handle's streams (steps AND episode generation) + error-flag check */

/* State exchange in the reference's own encoding (parity-harness state injection; for checkpoints use mg_save_state below).
 *   grid : (N, W, H, 3) u8 in Grid.encode() layout (core/grid.py:244-268), decoded like Grid.decode (270-289)
 *   agent: (N, 8) i32 = {x, y, dir, carry_type, carry_color, step_count, reset_pending, mission_id}
 * Grid.encode() is lossy where the reference's objects hold more than (type, colour, state), and so is this pair: a key hidden in a box
 * reads back as a plain box (Box.encode, world_object.py:65-67, knows nothing of Box.contains), the sentence levels are refused
 * (instruction trees and object identities are not in the encoding), and the level words are re-derived from grid + mission id where
 * that is possible (tracked GoTo / PutNext positions, obstacle lists) and KEPT as the handle has them where it is not (OpenDoor with a
 * location description: which doors "the door on your left" meant at reset time). */
MG_API int mg_get_state(mg_env* env, uint8_t* grid, int32_t* agent);
MG_API int mg_set_state(mg_env* env, const uint8_t* grid, const int32_t* agent);
/* Lossless checkpoint of a live handle -- what pickling a reference env carries (tests/test_envs.py:185-196 test_pickle_env): the
 * grids as the library holds them (hidden box contents included), agent records, level words (tracked positions, obstacle lists),
 * every env's np_random position, the sentence levels' instruction trees and object identities, the device policy's step counter.
 * mg_state_size bytes; mg_load_state needs a handle created with the same level / grid / batch size (any observation mode) and
 * re-draws the spare episodes from the restored stream positions.  Both synchronise. */
MG_API int mg_state_size(mg_env* env, int64_t* bytes);
MG_API int mg_save_state(mg_env* env, void* buf, int64_t bytes);
MG_API int mg_load_state(mg_env* env, const void* buf, int64_t bytes);

/* Per-env generator state (N, 5) u64 = {state_hi, state_lo, inc_hi, inc_lo, (has_uint32 << 32) | uinteger}:
 * numpy's PCG64 state as the reference env would hold it at this point of the episode sequence. */
MG_API int mg_get_rng(mg_env* env, uint64_t* out);
MG_API int mg_set_rng(mg_env* env, const uint64_t* in);

/* HIP-event timing on the handle's stream (bench.py uses these for the roofline line). */
MG_API int mg_timer_start(mg_env* env);
MG_API int mg_timer_stop(mg_env* env, float* elapsed_ms);   /* synchronises on the stop event */

/* counters since create: [0] env-steps executed, [1] episodes finished, [2] maps generated, [3] generator retries
 * (summed on the host from per-workgroup slots; synchronises the stream) */
MG_API int mg_get_counters(mg_env* env, uint64_t out[4]);

/* the EFFECTIVE depth R of the spare-episode ring (pre-generated next episodes per env): mg_config.spare_ring when given, else the level's default,
 * halved while the ring would exceed min(32 GB, a quarter of the device memory free at mg_create) -- so identical configs can differ from run to
 * run; a seeded mg_reset issues R + 1 generator launches.  1 for levels whose reset draws nothing; 0 = no ring (DynamicObstacles redraws in place).
 * (No reference counterpart: gymnasium resets draw inside reset(), minigrid_env.py:119-157.) */
MG_API int mg_ring_depth(mg_env* env);

MG_API const char* mg_last_error(mg_env* env);   /* env may be NULL for creation errors */
MG_API int mg_abi_version(void);
/* compile-time switches of this build as "key=value;..." ("attribution=0" in the product library: the MG_EXP step-skipping aid of
 * the attribution build is compiled out; bench.py prints the string into its JSON line next to every MG_* environment variable) */
MG_API const char* mg_build_info(void);
MG_API int mg_device_count(void);

/* Host-side self-test hooks (no GPU needed): run the library's own inline helpers on the CPU so that the
 * bit-parallel formulations can be checked exhaustively in the CPU test-suite. */
MG_API int mg_selftest_vis_row(uint32_t mask_in, uint32_t transparent, uint32_t* mask_out, uint32_t* up_out);
MG_API int mg_selftest_vis_row_n(int32_t view, uint32_t mask_in, uint32_t transparent, uint32_t* mask_out, uint32_t* up_out);
MG_API int mg_selftest_reward_lut(int32_t max_steps, double* out /* [max_steps+1] */);
/* k_step's observation stream packer (StreamEmit, mg_kernels.h) run lane by lane on the host: `in` = nenv x obe bytes,
 * each env's bytes split over lanes_per_env lanes as in the kernel and handed over as little-endian dwords; `out` must
 * reproduce them as one contiguous stream. */
MG_API int mg_selftest_stream(int32_t obe, int32_t nenv, int32_t lanes_per_env, const uint8_t* in, uint8_t* out);
/* Grid.render_tile (core/grid.py:145-198) for every tile of the RGB atlas, as the library renders it at mg_create:
 * out[51][5][2][tile_size][tile_size][3] = [tile key][no agent, agent_dir 0..3][plain, highlighted]; tile keys are
 * empty 0 | wall 1+c | floor 7+c | key 13+c | ball 19+c | box 25+c | door 31+3c+state | goal 49 | lava 50. */
MG_API int mg_render_tiles(int32_t tile_size, uint8_t* out);
MG_API int mg_selftest_pack_cell(int32_t type, int32_t color, int32_t state, uint32_t* code, uint32_t* triple);
/* k_roll7's observation pipeline (minigrid_amd/csrc/mg_roll.h) -- gen_obs_grid / process_vis / Grid.encode of the default 7x7 view
 * (minigrid_env.py:597-650, core/grid.py:244-328) as line gathers, byte transposes, carry-propagation visibility rows and the
 * output-space encode -- run on the host over states in mg_set_state's exchange format: grid (n, W, H, 3) u8, agent (n, 8) i32;
 * out (n, 7, 7, 3) u8.  mg_selftest_vis_row_carry is its process_vis row; mg_selftest_prims evaluates its VALU primitives
 * (perm / dot4 / bit reverse / bit-to-byte expand / visibility row / SDWA byte index) on the host or, on_device = 1, on the GPU: out[6][n]. */
MG_API int mg_selftest_obs7(int32_t width, int32_t height, int32_t n, const uint8_t* grid, const int32_t* agent, int32_t see_through,
                            uint8_t* out);
/* ... and FullyObsWrapper.observation (wrappers.py:419-426) as k_roll7<., true> produces it (the image-order code stream, the agent's cell, the same
 * output-space encode): grid (n, W, H, 3) u8, agent (n, 8) i32 -> out (n, W, H, 3) u8. */
MG_API int mg_selftest_obs_full(int32_t width, int32_t height, int32_t n, const uint8_t* grid, const int32_t* agent, uint8_t* out);
MG_API int mg_selftest_vis_row_carry(uint32_t mask_in, uint32_t transparent, uint32_t* mask_out, uint32_t* up_out);
MG_API int mg_selftest_prims(int32_t n, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* out, int32_t on_device);
/* DynamicObstaclesEnv's stream draws as the fused step kernel runs them per lane (minigrid_amd/csrc/mg_dynobs.h; reference:
 * envs/dynamicobstacles.py:110-157 + MiniGridEnv.place_obj / place_agent, minigrid_env.py:313-395), on the host, for n envs in the state
 * exchange format: grid (n, W, H, 3) u8 in/out, agent (n, 8) i32 in/out (x, y, dir), rng (n, 5) u64 in/out (mg_get_rng's words), obst (n) u64
 * in/out (byte i = cell index y * W + x of obstacle i, list order).  mode[i]: 0 = nothing, 1 = the obstacle moves of one step(), 2 = reset()
 * (agent_start = (sx, sy, sdir), sx < 0: place_agent).  flags[i]: bit 0 a placement failed, bit 1 the grid changed, bit 2 not_clear.
 * philox: 0 = numpy PCG64 streams, 1 = Philox streams, 2 = PCG64 with every try taken through the rare-case (redo) path of the draw code. */
/* MiniGridEnv.reset's _gen_grid of every level (envs/<level>.py _gen_grid, e.g. empty.py:97-114, doorkey.py:74-99, crossing.py:131-188, multiroom.py:118-300;
 * core/roomgrid.py; envs/babyai/<family>.py gen_mission + core/levelgen.py:24-211) as the lane-per-episode generator kernels run it on the env's numpy PCG64
 * stream (minigrid_amd/csrc/mg_genlane.h generate_one_lane, mg_gen.h), on the host: n envs seeded like reset(seed = seeds[i]), `episodes`
 * consecutive episodes each, drawn into host arrays laid out like the device's spare ring (one ring slot per episode) and read back:
 * grid (episodes, n, W, H, 3) u8, agent (episodes, n, 8) i32 (x, y, dir, carried type, carried colour, record flags, step count, mission id),
 * aux (episodes, n) u64, rng (episodes, n, 5) u64 = the stream after each episode (mg_get_rng's words), failed (episodes, n) u8,
 * instr: NULL or (episodes, n, 40) u64 (the sentence levels' instruction record).  The product library's lane kernels serve the single-room
 * levels; the others' per-lane forms run on the device in the MG_LANE_WIDE build (mg_genlane.h).  DynamicObstacles: mg_selftest_dynobs. */
MG_API int mg_selftest_generate(const mg_config* cfg, int32_t n, int32_t episodes, const uint64_t* seeds, uint8_t* grid, int32_t* agent, uint64_t* aux,
                                uint64_t* rng, uint8_t* failed, uint64_t* instr);
/* MiniGridEnv.step (minigrid_env.py:525-595) + the level's own step rule (envs/<level>.py, e.g. fetch.py:162-175, unlock.py:90-98) as the step kernels run
 * them per lane (minigrid_amd/csrc/mg_step.h env_transition), on the host: ONE step of n independent envs, no autoreset, state exchange format
 * (grid (n, W, H, 3) u8 and agent (n, 8) i32 in / out: x, y, dir, carried type, carried colour, step count, -, mission id).  group / rule / rule_cell /
 * rule_div = the kernel variant and level rule mg_create derives from the config (enum values in mg_step.h).  aux: NULL, or -- the single-room BabyAI
 * GoTo levels, grids of at most 64 cells -- (n, 2) u64 in / out: GoToInstr's tracked positions and where the described objects are now (bit
 * y * W + x).  The other rules that keep an auxiliary word per env (GoToObject, PutNear, the multi-room GoTo / PutNext / OpenDoor levels) and the
 * sentence levels are refused. */
MG_API int mg_selftest_transition(int32_t group, int32_t rule, int32_t rule_cell, int32_t rule_div, int32_t width, int32_t height, int32_t max_steps,
                                  int32_t no_death_mask, double death_cost, int32_t n, uint8_t* grid, int32_t* agent, const uint8_t* actions,
                                  double* reward, uint8_t* terminated, uint8_t* truncated, uint32_t* errbits, uint64_t* aux);
/* RoomGridLevel.step's second half for the levels whose mission is an instruction tree (envs/babyai/core/roomgrid_level.py:87-104,
 * verifier.py:228-571: update_objs_poss, ActionInstr.verify incl. use_done_actions, And / Before / After, object identity through pickup / drop /
 * Box.toggle) as the step kernels run it per lane (minigrid_amd/csrc/mg_verify.h), on the host, for n independent cases: grid (n, W, H, 3) u8 and
 * agent (n, 8) i32 = the state AFTER the action (state exchange format), actions (n) u8, records (n, 40) u64 in / out (the instruction record:
 * minigrid_amd/csrc/mg_device.h); status (n) i32 = 0 continue | 1 success | 2 failure, max_steps (n) i32, errbits (n) u32 (8 = tracking error). */
MG_API int mg_selftest_verify(int32_t width, int32_t height, int32_t n, int32_t done_actions, const uint8_t* grid, const int32_t* agent,
                              const uint8_t* actions, uint64_t* records, int32_t* status, int32_t* max_steps, uint32_t* errbits);
MG_API int mg_selftest_dynobs(int32_t width, int32_t height, int32_t n_obstacles, int32_t sx, int32_t sy, int32_t sdir, int32_t philox, int32_t n,
                              const uint8_t* mode, uint8_t* grid, int32_t* agent, uint64_t* rng, uint64_t* obst, uint8_t* flags);

#ifdef __cplusplus
}
#endif
#endif /* MINIGRID_HIP_H */
