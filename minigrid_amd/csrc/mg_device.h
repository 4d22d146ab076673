// mg_device.h — cell encoding, object predicates and the bit-parallel visibility row shared by the kernels.
// Product code (gfx950).  Reference semantics cited per function (paths relative to the reference root).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MG_HD __host__ __device__ __forceinline__
#define MG_D __device__ __forceinline__

namespace mg {

// ---- object / colour / state codes: core/constants.py:20,25-37,42-46 ----
enum : uint32_t { T_UNSEEN = 0, T_EMPTY = 1, T_WALL = 2, T_FLOOR = 3, T_DOOR = 4, T_KEY = 5, T_BALL = 6,
                  T_BOX = 7, T_GOAL = 8, T_LAVA = 9, T_AGENT = 10,
                  // internal-only type codes so a cell fits one byte: a door's state rides in the type nibble
                  T_DOOR_CLOSED = 11, T_DOOR_LOCKED = 12 };
enum : uint32_t { C_RED = 0, C_GREEN = 1, C_BLUE = 2, C_PURPLE = 3, C_YELLOW = 4, C_GREY = 5 };
// core/actions.py:7-20
enum : uint32_t { A_LEFT = 0, A_RIGHT = 1, A_FORWARD = 2, A_PICKUP = 3, A_DROP = 4, A_TOGGLE = 5, A_DONE = 6 };

// A grid cell in HBM/LDS is ONE byte: code = type | colour << 4 | opaque << 7.
//   empty (Python None)  -> T_EMPTY, colour 0  == 0x01, which decodes to the reference's (1,0,0) for free
//   door open/closed/locked -> type 4 / 11 / 12 (state 0 / 1 / 2)
//   bit 7 = "NOT see_behind()" (world_object.py:164,181-182: wall, closed door, locked door), kept in the code so
//           that the visibility pass reads it with one bit-field extract; make_cell() is the only place that sets it
//   0x00 is "no object" for the carrying slot.
//   type 13 (internal) marks the agent's own cell in the FullyObs encode: colour field = agent_dir.
//   type 14 (internal) is a GREY box whose `contains` is a key (Box.contains, world_object.py:272-293; ObstructedMaze hides
//           every key in a box of colour COLOR_NAMES[2] = grey, obstructedmaze.py:120,161-164): colour field = the KEY's colour.
//           It encodes, renders and is picked up as the grey box; toggling it leaves the key in its place.
//   type 15 (internal) is a box of colour c holding the key of the level's ONLY door (BabyAI KeyInBox, unlock.py:232-242): the
//           key's colour is the door's, found on the grid when the box is opened (k_step).
constexpr uint32_t T_AGENT_MARK = 13;
constexpr uint32_t T_BOX_KEY = 14;
constexpr uint32_t T_BOX_DOORKEY = 15;
constexpr uint32_t OPAQUE_TYPES = (1u << T_WALL) | (1u << 11) | (1u << 12);
constexpr uint32_t OPAQUE_BIT = 0x80u;

MG_HD uint32_t cell_type(uint32_t code) { return code & 15u; }
MG_HD uint32_t cell_color(uint32_t code) { return (code >> 4) & 7u; }
MG_HD constexpr uint32_t make_cell(uint32_t type, uint32_t color) {
  return type | (color << 4) | (((OPAQUE_TYPES >> type) & 1u) << 7);
}
constexpr uint32_t CELL_EMPTY = T_EMPTY;
constexpr uint32_t CELL_WALL_GREY = make_cell(T_WALL, C_GREY);
constexpr uint32_t CELL_GOAL = make_cell(T_GOAL, C_GREEN);
constexpr uint32_t CELL_LAVA = make_cell(T_LAVA, C_RED);
constexpr uint32_t CELL_BALL_RED = make_cell(T_BALL, C_RED);
constexpr uint32_t CELL_BALL_BLUE = make_cell(T_BALL, C_BLUE);

// (type, colour, state) of the reference encoding -> cell code.  Mirrors WorldObj.decode (core/world_object.py:69-102):
// empty/unseen/agent -> None; Goal()/Lava() take their default colours; non-door state is ignored.
MG_HD uint32_t cell_from_triple(uint32_t type, uint32_t color, uint32_t state) {
  if (type == T_EMPTY || type == T_UNSEEN || type >= T_AGENT) return CELL_EMPTY;
  if (type == T_GOAL) return CELL_GOAL;
  if (type == T_LAVA) return CELL_LAVA;
  if (type == T_DOOR) type = state == 0 ? (uint32_t)T_DOOR : (state == 2 ? (uint32_t)T_DOOR_LOCKED : (uint32_t)T_DOOR_CLOSED);
  return make_cell(type, color & 7u);
}
// cell code -> type | colour << 8 | state << 16, i.e. the three bytes WorldObj.encode / Door.encode produce
// (core/world_object.py:65-67,196-212) and Grid.encode writes for None (core/grid.py:260-263).
MG_HD uint32_t cell_triple(uint32_t code) {
  uint32_t t = code & 15u, c = (code >> 4) & 7u;
  if (t == T_AGENT_MARK) return (uint32_t)T_AGENT | ((uint32_t)C_RED << 8) | (c << 16);   // wrappers.py:422-424
  if (t == T_BOX_KEY) return (uint32_t)T_BOX | ((uint32_t)C_GREY << 8);
  if (t == T_BOX_DOORKEY) return (uint32_t)T_BOX | (c << 8);
  uint32_t st = t >= T_DOOR_CLOSED ? t - 10u : 0u;
  t = t >= T_DOOR_CLOSED ? (uint32_t)T_DOOR : t;
  return t | (c << 8) | (st << 16);
}

// predicate bitmaps over the 4-bit type code
// can_overlap (world_object.py:113,128,141,177-179): goal, floor, lava, OPEN door; None (empty) is walkable too
constexpr uint32_t WALKABLE_MASK = (1u << T_EMPTY) | (1u << T_FLOOR) | (1u << T_DOOR) | (1u << T_GOAL) | (1u << T_LAVA);
// can_pickup (world_object.py:243,265,277)
constexpr uint32_t PICKUP_MASK = (1u << T_KEY) | (1u << T_BALL) | (1u << T_BOX) | (1u << T_BOX_KEY) | (1u << T_BOX_DOORKEY);

MG_HD bool cell_walkable(uint32_t code) { return (WALKABLE_MASK >> (code & 15u)) & 1u; }
MG_HD bool cell_pickable(uint32_t code) { return (PICKUP_MASK >> (code & 15u)) & 1u; }
MG_HD bool cell_transparent(uint32_t code) { return !(code & OPAQUE_BIT); }

// Door.toggle (world_object.py:184-194) / Box.toggle (290-293, contains is None on this path) on a cell code.
// Returns the new code (unchanged when toggling does nothing).
MG_HD uint32_t cell_toggle(uint32_t code, uint32_t carry) {
  const uint32_t t = code & 15u, col = (code >> 4) & 7u;
  if (t == T_DOOR_LOCKED) {
    bool has_key = (carry & 15u) == T_KEY && ((carry >> 4) & 7u) == col;
    return has_key ? make_cell(T_DOOR, col) : code;
  }
  if (t == T_DOOR) return make_cell(T_DOOR_CLOSED, col);
  if (t == T_DOOR_CLOSED) return make_cell(T_DOOR, col);
  if (t == T_BOX) return CELL_EMPTY;
  if (t == T_BOX_KEY) return make_cell(T_KEY, col);       // Box.toggle: replaced by what it contains (world_object.py:290-293)
  return code;
}

// ---- one row of Grid.process_vis (core/grid.py:291-328), bit-parallel ----
// Row j of the 7-wide view: `m` = mask bits already set in this row (bit i = mask[i][j]), `t` = transparency bits
// (cell is None or see_behind()).  The reference sweeps i = 0..5 left-to-right (lit & transparent cell lights
// i+1 in this row and i, i+1 in row j-1), then i = 6..1 right-to-left (lights i-1 / i-1, i above), each sweep
// seeing its own writes.  A sweep is a one-directional occluded fill, done here with Kogge-Stone steps.
// Returns the final row mask in *m_out and the bits contributed to row j-1 in *up_out.
// tests/test_abi_cpu.py (test_vis_row_bit_parallel_equals_reference_loops_exhaustively) checks all 2^14 inputs against the literal loops.
MG_HD void vis_row(uint32_t m, uint32_t t, uint32_t* m_out, uint32_t* up_out) {
  // Both sweeps start from the same seeds: a cell lit by sweep 1 is reached through transparent cells from a seed
  // s, so sweeping left from it only re-walks the run back to s and then continues as s itself would.  Hence the
  // lit-and-transparent set after both sweeps is fill_right(m&t) | fill_left(m&t), two independent occluded fills.
  const uint32_t g0 = m & t;
  uint32_t gr = g0, pr = t, gl = g0, pl = t;
  gr |= pr & (gr << 1); pr &= pr << 1;      gl |= pl & (gl >> 1); pl &= pl >> 1;
  gr |= pr & (gr << 2); pr &= pr << 2;      gl |= pl & (gl >> 2); pl &= pl >> 2;
  gr |= pr & (gr << 4);                     gl |= pl & (gl >> 4);
  const uint32_t s1 = gr & 0x3Fu;           // sweep-1 sources i = 0..5: light i+1 here, i and i+1 above
  const uint32_t s2 = (gr | gl) & 0x7Eu;    // sweep-2 sources i = 6..1: light i-1 here, i and i-1 above
  *m_out = (m | (s1 << 1) | (s2 >> 1)) & 0x7Fu;
  *up_out = (s1 | (s1 << 1) | s2 | (s2 >> 1)) & 0x7Fu;
}

// The same for a view of width V <= 16 (ViewSizeWrapper, wrappers.py:629-673): one more Kogge-Stone step, masks from V.
MG_HD void vis_row_n(uint32_t m, uint32_t t, int V, uint32_t* m_out, uint32_t* up_out) {
  const uint32_t full = (1u << V) - 1u;
  const uint32_t g0 = m & t;
  uint32_t gr = g0, pr = t, gl = g0, pl = t;
  gr |= pr & (gr << 1); pr &= pr << 1;      gl |= pl & (gl >> 1); pl &= pl >> 1;
  gr |= pr & (gr << 2); pr &= pr << 2;      gl |= pl & (gl >> 2); pl &= pl >> 2;
  gr |= pr & (gr << 4); pr &= pr << 4;      gl |= pl & (gl >> 4); pl &= pl >> 4;
  gr |= pr & (gr << 8);                     gl |= pl & (gl >> 8);
  const uint32_t s1 = gr & (full >> 1);     // sweep-1 sources i = 0..V-2
  const uint32_t s2 = (gr | gl) & full & ~1u;   // sweep-2 sources i = V-1..1
  *m_out = (m | (s1 << 1) | (s2 >> 1)) & full;
  *up_out = (s1 | (s1 << 1) | s2 | (s2 >> 1)) & full;
}

// COLOR_NAMES is sorted alphabetically (core/constants.py:17): blue, green, grey, purple, red, yellow -> COLOR_TO_IDX
MG_HD uint32_t color_from_sorted(uint32_t i) {
  const uint32_t packed = (C_BLUE) | (C_GREEN << 4) | (C_GREY << 8) | (C_PURPLE << 12) | (C_RED << 16) | (C_YELLOW << 20);
  return (packed >> (4u * i)) & 15u;
}

// reference OBJECT_TO_IDX of a cell code (the internal closed/locked door types are doors)
MG_HD uint32_t cell_ref_type(uint32_t code) { const uint32_t t = code & 15u; return (t == T_BOX_KEY || t == T_BOX_DOORKEY) ? (uint32_t)T_BOX : (t >= T_DOOR_CLOSED ? (uint32_t)T_DOOR : t); }

// agent record: one u64 per env
//   byte 0 x, 1 y, 2 dir (bits 0-1) | mission id bits 8-13 (bits 2-7), 3 carrying (cell code, 0 = nothing), 4-5 step_count (u16),
//   6 flags, 7 mission id bits 0-7.  Mission ids are 14 bits: PutNear has 324 missions (putnear.py:72-80).
constexpr uint32_t FLAG_RESET_PENDING = 1u;   // previous step ended the episode; NEXT_STEP autoreset is due
// levels whose env stream is also consumed by step() (DynamicObstacles) cannot pre-draw a spare episode: their resets
// are drawn by a generator launch right before the step launch, which then only observes the fresh episode
constexpr uint32_t FLAG_FRESH = 2u;           // regenerated just before this launch: observe, do not step
constexpr uint32_t FLAG_NOT_CLEAR = 4u;       // DynamicObstacles: the front cell was occupied before the obstacles moved
constexpr uint32_t FLAG_TARGETS_STALE = 8u;   // BabyAI GoTo levels: a described object moved since GoToInstr's positions were refreshed
constexpr uint32_t FLAG_SHOW_TAKEN = 16u;     // PutNext(start_carrying): the episode's first observation shows the carried object where it was taken from
constexpr uint32_t FLAG_NEW_EPISODE = 32u;    // sentence levels: k_step took a spare episode; k_verify installs its instruction record
// spare episodes only: drawing this episode met RoomGrid.place_agent's endless loop (mg_gen.h room_stuck) -- the reference would never
// return from the reset() that reaches it.  Taking the episode out of the ring reports ERR_GENERATOR (take_spare, mg_step.h).
constexpr uint32_t FLAG_STUCK = 64u;
// BabyAI levels with ONE action instruction under use_done_actions (verifier.py:26, 222-242): ActionInstr.lastStepMatch -- the previous
// action completed the instruction.  Cleared with the other flags when an episode is taken (a fresh instruction per mission).
constexpr uint32_t FLAG_LAST_MATCH = 128u;
struct Agent {
  uint32_t x, y, dir, carry, step, flags, mission;
};
MG_HD Agent agent_unpack(uint64_t r) {
  Agent a;
  a.x = (uint32_t)(r & 0xFF); a.y = (uint32_t)((r >> 8) & 0xFF); a.dir = (uint32_t)((r >> 16) & 3u);
  a.carry = (uint32_t)((r >> 24) & 0xFF); a.step = (uint32_t)((r >> 32) & 0xFFFF);
  a.flags = (uint32_t)((r >> 48) & 0xFF); a.mission = (uint32_t)((r >> 56) & 0xFF) | ((uint32_t)((r >> 18) & 0x3Fu) << 8);
  return a;
}
MG_HD uint64_t agent_pack(const Agent& a) {
  return (uint64_t)(a.x & 0xFF) | ((uint64_t)(a.y & 0xFF) << 8) | ((uint64_t)((a.dir & 3u) | (((a.mission >> 8) & 0x3Fu) << 2)) << 16) |
         ((uint64_t)(a.carry & 0xFF) << 24) | ((uint64_t)(a.step & 0xFFFF) << 32) |
         ((uint64_t)(a.flags & 0xFF) << 48) | ((uint64_t)(a.mission & 0xFF) << 56);
}

// ---- the general BabyAI instruction as data (envs/babyai/core/verifier.py), INSTR_WORDS u64 per env ("sentence levels") ----
// Up to four action instructions (leaves 0..3) under And / Before / After nodes (4..6): LevelGen's grammar (levelgen.py:157-211)
// never nests deeper than Before/After(And(l, l), And(l, l)).  Objects are identified by an id (0..62) given at generation; the
// record tracks where every object is (`pos`), which objects each description selected at reset (ObjDesc.obj_set), and -- for
// GoToInstr / PutNextInstr, whose obj_poss are POSITIONS refreshed only at reset and by drop actions -- the cells a member left
// since the last refresh (`stale`, see RULE_GOTO_BIG for the argument).
//   word 0       header: [0:3) root | [3:27) 3 nodes x (kind 2: 1 Before 2 After 3 And | a 3 | b 3) | [27:39) 3 x (a_done 2 | b_done 2)
//                | [39:55) this episode's max_steps | [55:62) id + 1 of the carried object (0 = none)
//   words 1..4   leaf k: [0:20) leaf20 = verb 2 (go to, pick up, open, put next) | desc 9 | fixed desc 9 (PutNext) | [20] strict
//                | [21:28) preCarrying id + 1;  desc 9 = type 2 (door key ball box) | colour 3 (0 any, COLOR_TO_IDX + 1) | loc 3
//                (0 none, left right front behind) | article 1 (more than one object matched at reset)
//   words 5..12  obj_set of leaf k's desc (5 + 2k) and fixed desc (6 + 2k), bit = id
//   words 13..20 stale cells of the same descriptions: four u16 cell indices each (0xFFFF = free)
//   words 21..36 pos: u16 cell index y * W + x per id; 0xFFFF carried, 0xFFFE gone (a toggled box)
//   words 37, 38 the mission as data (what the host turns into the sentence, Instr.surface): [0:60) leaf20 of leaves 0..2 |
//                [60:63) root;  [0:20) leaf20 of leaf 3 | [20:44) the three nodes
constexpr int INSTR_WORDS = 40;
constexpr int IW_LEAF = 1, IW_SET = 5, IW_STALE = 13, IW_POS = 21, IW_MISSION = 37;
constexpr uint32_t POS_CARRIED = 0xFFFFu, POS_GONE = 0xFFFEu;
enum : uint32_t { V_GOTO = 0, V_PICKUP = 1, V_OPEN = 2, V_PUTNEXT = 3 };
enum : uint32_t { N_BEFORE = 1, N_AFTER = 2, N_AND = 3 };
enum : uint32_t { R_CONTINUE = 0, R_SUCCESS = 1, R_FAILURE = 2 };
MG_HD constexpr uint32_t desc9(uint32_t ref_type, uint32_t color_plus1, uint32_t loc, uint32_t article) {
  return (ref_type - T_DOOR) | (color_plus1 << 2) | (loc << 5) | (article << 8);
}
MG_HD constexpr uint32_t desc9_type(uint32_t d) { return (d & 3u) + T_DOOR; }
MG_HD constexpr uint32_t desc9_color(uint32_t d) { return (d >> 2) & 7u; }
MG_HD constexpr uint32_t desc9_loc(uint32_t d) { return (d >> 5) & 7u; }
MG_HD constexpr uint32_t leaf20(uint32_t verb, uint32_t d, uint32_t f) { return verb | (d << 2) | (f << 11); }
MG_HD constexpr uint32_t node8(uint32_t kind, uint32_t a, uint32_t b) { return kind | (a << 2) | (b << 5); }

// DIR_TO_VEC (core/constants.py:49-58) without a table: dir 0:(1,0) 1:(0,1) 2:(-1,0) 3:(0,-1)
MG_HD int dir_dx(uint32_t d) { return (d & 1u) ? 0 : 1 - (int)(d & 2u); }
MG_HD int dir_dy(uint32_t d) { return (d & 1u) ? 1 - (int)(d & 2u) : 0; }

// device error kinds (surfaced by mg_sync / mg_copy_outputs).  The flags live in mapped pinned HOST memory, one word per kind
// (word k = bit k), set with plain stores of 1 -- no device atomics on host memory, no device-to-host copy to read them.
enum : uint32_t { ERR_BAD_ACTION = 1u, ERR_GENERATOR = 2u, ERR_OOB = 4u, ERR_TRACKED = 8u };
constexpr int ERR_WORDS = 6;                  // the four kinds above + mg_set_state's "bad agent record" + [5] an inter-wave spin of k_roll7 ran past
                                              // MG_SPIN_BOUND polls (only -DMG_SPIN_BOUND builds ever set it: mg_roll.h)
constexpr int ERR_WORD_SPIN = 5;
MG_HD void report_errors(uint32_t* err, uint32_t bits) {
#pragma unroll
  for (int k = 0; k < 4; k++) if ((bits >> k) & 1u) err[k] = 1u;
}

// generator / rule groups (see mg_gen.h "Groups")
enum : int { GG_NONE = 0, GG_LIGHT = 1, GG_ROOMGRID = 2, GG_ROOMS = 4, GG_SENTENCE = 8, GG_ALL = 15,
              GG_DYNOBS = 16 };    // (step kernels only: k_roll7 with DynamicObstacles' moves and resets inside the step loop, mg_dynobs.h)
// A STEP kernel's GG may also name ONE rule of its group (round 6): the group in the low byte, rule + 1 above it.  gg_group / gg_rule take it apart; the generators
// only ever see plain groups.
constexpr int GG_RULE(int group, int rule) { return group | ((rule + 1) << 8); }
MG_HD constexpr int gg_group(int GG) { return GG & 0xFF; }
MG_HD constexpr int gg_rule(int GG) { return (GG >> 8) - 1; }          // -1: the launch parameter decides (StepParams::rule)
MG_HD int gen_group_of_kind(int kind) { return (kind >= 50 && kind <= 53) ? GG_SENTENCE : (kind >= 21 && kind <= 49) ? GG_ROOMS : (kind == 3 || (kind >= 9 && kind <= 11) || kind == 14 || (kind >= 16 && kind <= 20)) ? GG_ROOMGRID : GG_LIGHT; }

// `counters` layout (u64): [0..15] scratch (debug stamps) | one episodes-finished slot per 64-env wave |
// STAT_GEN_SLOTS x {maps generated, whole-map retries}; mg_get_counters sums them on the host
constexpr int STAT_EPISODES = 16;
constexpr uint32_t STAT_GEN_SLOTS = 4096;

}  // namespace mg
