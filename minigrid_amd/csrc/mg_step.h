// mg_step.h — HIP kernels for gfx950 (MI355X): the lockstep MiniGridEnv.step()/gen_obs()/FullyObs hot path as a
// T-step fused rollout kernel, the asynchronous episode generator that keeps a ring of spare episodes per env full,
// and seeding.  One wavefront lane per environment; one wavefront = 64 consecutive envs = one workgroup.
//
// HBM layout (all per mg_env handle; N envs, env-major):
//   grid        u8  [N][CS]      one byte per cell (mg_device.h), row-major y*W+x, CS = W*H rounded up to 16
//   spare_grid  u8  [R][N][CS]   ring of R pre-generated NEXT episodes per env (R = 1 for levels that draw nothing)
//   agent       u64 [N]          packed agent record (x, y, dir, carrying, step_count, flags, mission)
//   spare_agent u64 [R][N], spare_aux u64 [R][N]
//   head / tail u32 [N]          spares consumed / generated so far; slot = count & (R-1); after a flush tail = head + R
//   rng u64 [5][N]               generator state after the last generated spare; rng_snap u64 [R][5][N] = as it was
//                                before slot s was drawn (what the reference env's np_random would hold at that point)
//   out         [S] slots of { obs u8 [N][obe] | reward f64 [N] | terminated, truncated, direction, mission, action u8 [N] }:
//               the trajectory ring; slot 0 is always the most recent step (mg_get_outputs), slot k the step k calls ago
//
// Why spare episodes: in the reference an env's np_random stream is consumed ONLY by reset() on this path, so the
// maps of episodes k+1, k+2, ... can be drawn any time after episode k's map without changing the stream.  The step
// kernel therefore never runs a generator: on (auto)reset it takes the next spare out of the ring and files a refill
// request; a generator kernel on a SECOND stream (one wavefront per request, mg_gen.h) refills the ring while later
// step launches run.  The host orders the two streams with events so that a slot is never consumed before its refill
// completed (mg_api.hip: batches); the sequential PCG64 + rejection-sampling code is off the step critical path.
#pragma once
#include "mg_device.h"
#include "mg_tiles.h"
#include "mg_rng.h"

// attribution aid: -DMG_ATTRIBUTION builds (profiles/attr_build.py, a SEPARATE library selected with MINIGRID_AMD_LIB) read MG_EXP and
// skip parts of a step so that their cost can be timed.  The product library is built without it: the switch folds to 0 and a stray
// MG_EXP in somebody's environment cannot make it produce garbage.
#if defined(MG_ATTRIBUTION)
#define MG_EXPBIT(P, b) (((P).exp & (b)) != 0)
#else
#define MG_EXPBIT(P, b) false
#endif

namespace mg {

constexpr int VIEW = 7;
constexpr int VIEW_CELLS = VIEW * VIEW;        // 49
constexpr int PARTIAL_OBS_BYTES = VIEW_CELLS * 3;  // 147
constexpr int MAX_FUSED_STEPS = 32;                // steps per k_step launch: the LDS action staging [T][EPW] is sized for this

enum : int { PHASE_STEP = 0, PHASE_OBSERVE = 1 };
enum : int { ACT_SRC_BUFFER = 0, ACT_SRC_PHILOX = 1 };
// The level rule as a kernel reads it.  A step kernel is instantiated per rule GROUP (GG: it carries its group's rules behind wave-uniform tests of the launch parameter)
// or -- round 6 -- for ONE rule of a group (GG_RULE(group, rule), mg_device.h: the rule is a compile-time constant and the group's other rules are compiled out;
// carrying them cost the dynamics loop 6-12 %: profiles/r6/ab_fixed_rule_*.txt).
#define MG_RULE(GG, P) (gg_rule(GG) >= 0 ? gg_rule(GG) : (P).rule)
enum : int { RULE_NONE = 0, RULE_GOTO = 1, RULE_FETCH = 2, RULE_GOTODOOR = 3, RULE_UNLOCK = 4, RULE_PICKUP = 5,
              RULE_REDBLUE = 6, RULE_MEMORY = 7, RULE_DYNOBS = 8, RULE_GOTOOBJ = 9,
              RULE_PICKUPDESC = 10, RULE_OPENFRONT = 11, RULE_PUTNEAR = 12, RULE_GOTO_BIG = 13, RULE_PUTNEXT = 14, RULE_OPENDOOR = 15, RULE_SENTENCE = 16 };

struct StepParams {
  // ---- state ----
  uint8_t* grid; uint64_t* agent;
  uint64_t* aux;                                  // BabyAI GoTo levels: bitboard of the tracked target positions
  const uint8_t* spare_grid; const uint64_t* spare_agent; const uint64_t* spare_aux;   // rings [R][N]...
  uint32_t* head; uint32_t ring_mask;             // spares consumed per env; slot = head & ring_mask
  uint32_t* seg; uint32_t* seg_count; int seg_cap;   // this batch's refill requests: one segment of seg_cap env ids per wave
  // ---- inputs ----
  const void* actions; int act_dtype; int act_src; uint64_t action_seed; uint32_t t0;   // buffer: [T][N] of act_dtype
  int staged;                                     // k_roll7: 1 = the STAGED instantiation (big grids: one copy of the grids per workgroup, the dynamics wave stages the codes; mg_roll.h)
  const uint8_t* obs_mask;                        // PHASE_OBSERVE of a masked reset(): only these envs take a new episode
  uint64_t* instr; const uint64_t* spare_instr; unsigned long long off_sentence;   // sentence levels through k_roll7<GG_SENTENCE>: the verifier runs inside the step loop
  // ---- outputs: slot s of the trajectory ring starts at out + s * slot_bytes; step j of this launch -> slot slot0 - j (mod S) ----
  uint8_t* out; unsigned long long slot_bytes, off_reward, off_term, off_trunc, off_dir, off_mission, off_action;   // field offsets of env 0's mg_step_scalars (16 bytes per env)
  uint8_t* obs; unsigned long long obs_stride;    // observation stream of slot s at obs + s * obs_stride (= out / slot_bytes, or the RGB tile map)
  unsigned long long obs_wg_stride;               // k_roll7: ... + workgroup * obs_wg_stride (64 * OBE; S * 64 * OBE in the env-block-major trajectory layout)
  int T, slot0, S;
  // ---- tables / bookkeeping ----
  uint32_t* err; unsigned long long* counters;
  // ---- config ----
  int N, W, H, CS, GS, cells, max_steps, see_through, rule, rule_cell, rule_div, autoreset_next_step, autoreset_same_step, phase, static_gen;
  int live_gen;           // resets are drawn in place right before the step launch (DynamicObstacles): queue the ended envs (2: k_roll7<GG_DYNOBS> replaces the list, see launch_step)
  int use_shadow;         // spare episodes of every env staged in LDS (its shadow slots) at launch start: 0 (one-step launches), 1, or 2 (k_roll7)
  int shadow_stride, spr_stride;   // bytes between the shadow sets of the staged grids / of the staged (agent record, aux word) pairs
  int off_grid, off_shadow, off_spr, off_act, off_trow, off_T, OBE;   // LDS carve-up (bytes); OBE = obs bytes per env
  int view;               // agent view size V (odd, 3..15)
  int no_death_mask; double death_cost;   // NoDeath wrapper (wrappers.py:845-882)
  uint32_t cpe_magic;     // ceil(2^20 / (CS/16))
  int rgb_full, rgb_highlight;   // MODE 4 (tile map for k_render): whole grid + highlight mask instead of the agent's view
  long long env_base;
  int exp;                // tuning aid (MG_EXP, never set in normal use): k_roll7 skips parts of a step so that their cost can be timed
  int codes_stride, off_shadow_gt; uint32_t w_magic, h_magic;   // k_roll7: bytes between the waves' code streams; FullyObs: shadow image stream, ceil(2^16 / W), ceil(2^16 / H)
  int share;              // k_roll7, one-step launches: the workgroup's waves share the output-space encode of wave 0's step
  int split[5];           // k_roll7 (mg_roll.h): wave w of a workgroup produces steps [split[w], split[w + 1])
  int done_actions;          // BabyAI levels: verifier.py's use_done_actions (only the `done` action reports; mg_config.babyai_done_actions)
  int epw;                   // k_roll7: envs per workgroup (64, or 32 for small batches; mg_api.hip configure_obs)
  int nt;                    // k_roll7: observation stores are nontemporal (a long burst of launches; mg_api.hip launch_step)
  int split_mode, off_log;   // k_roll7: 1 = wave 0 runs the dynamics once and logs them (ring at off_log), the other waves encode
  // k_roll7<GG_DYNOBS>: DynamicObstacles with its stream draws inside the step loop (mg_dynobs.h)
  uint64_t* rng;             // the envs' streams (SoA words, mg_rng.h)
  int dyn_n, dyn_sx, dyn_sy, dyn_sdir;   // n_obstacles; agent_start_pos / agent_start_dir (dyn_sx < 0: place_agent)
  int off_tmpl;              // LDS: the level's constant grid (walls + goal), CS bytes
  int dring;                 // k_roll7<GG_DYNOBS / GG_SENTENCE>, split: code stagings in the ring between the dynamics wave and the encode waves (2 or 4)
  int off_instr;             // k_roll7<GG_SENTENCE>, LDS: the workgroup's instruction records (mg_roll.h ROLL_INSTR_STRIDE)
  int stat_gen_off;          // first generator statistics slot in `counters`
};

// _reward() = 1 - 0.9 * (step_count / max_steps), three separately rounded f64 ops (minigrid_env.py:240-245).
// Normally read from the host-built LUT; this exact device form covers step_count > max_steps (autoreset disabled).
// (host form, for mg_selftest_transition: the same three roundings -- the library is built with -ffp-contract=off, the volatiles keep the compiler
// from re-associating)
MG_HD double reward_exact(uint32_t step, int max_steps) {
#if defined(__HIP_DEVICE_COMPILE__)
  double q = __ddiv_rn((double)step, (double)max_steps);
  double p = __dmul_rn(0.9, q);
  return __dsub_rn(1.0, p);
#else
  volatile double q = (double)step / (double)max_steps;
  volatile double p = 0.9 * q;
  return 1.0 - p;
#endif
}
MG_HD double dadd_rn(double a, double b) {          // one IEEE-rounded addition
#if defined(__HIP_DEVICE_COMPILE__)
  return __dadd_rn(a, b);
#else
  volatile double r = a + b;
  return r;
#endif
}

// Uniform-random policy on the device: Philox4x32-10 keyed by action_seed, counter = (global env index, t / 4); the
// four output words are the actions of steps 4*(t/4) .. 4*(t/4)+3, each mapped to Discrete(7) (minigrid_env.py:63).
MG_D void philox_action_block(const StepParams& P, int e, uint32_t tblk, uint32_t w[4]) {
  const uint64_t gi = (uint64_t)(P.env_base + e);
  w[0] = (uint32_t)gi; w[1] = (uint32_t)(gi >> 32); w[2] = tblk; w[3] = 0x41435431u;
  philox4x32_10(w, (uint32_t)P.action_seed, (uint32_t)(P.action_seed >> 32));
}
MG_D uint32_t load_action(const StepParams& P, int e, int j) {      // prologue only: the loop reads the staged copy from LDS
  const size_t i = (size_t)j * (size_t)P.N + (size_t)e;
  if (P.act_dtype == 0) return ((const uint8_t*)P.actions)[i];
  if (P.act_dtype == 1) { const int32_t v = ((const int32_t*)P.actions)[i]; return (v < 0 || v > 255) ? 255u : (uint32_t)v; }
  const long long v = ((const long long*)P.actions)[i];
  return (v < 0 || v > 255) ? 255u : (uint32_t)v;
}

// bits k in [0,V-1] with 0 <= c0 + s*k < L (s = +1 or -1): the in-bounds run of a view row/column
MG_D uint32_t inb_mask_v(int c0, int s, int L, int V) {
  const int lo = s > 0 ? max(0, -c0) : max(0, c0 - (L - 1));
  const int hi = s > 0 ? min(V - 1, L - 1 - c0) : min(V - 1, c0);
  const uint32_t m = ((2u << (hi & 31)) - 1u) & ~((1u << (lo & 31)) - 1u);
  return hi >= lo ? m : 0u;
}

// ======================================================================================================
// Observation byte stream.  The 64 envs of a wave produce ONE contiguous byte stream (env-major, obe bytes per env) that
// is staged in LDS and copied out with 16 B per lane.  obe is odd (147, 243, ...), so an env's bytes do not start on a
// dword of the stream; byte stores at a 147-byte lane stride were the LDS-conflict hot spot of the previous kernel.
// Here every lane packs its env's bytes into dwords IN REGISTERS (D[0], D[1], ...: little-endian, stream order) and
// hands them to StreamEmit, which shifts them by the env's phase and writes only whole, aligned stream dwords:
//   B                               first stream byte of the lane's byte range (l * obe with one lane per env)
//   q = (-B) & 3                    leading bytes that belong to the last dword STARTING in env l-1
//   lane l writes the dwords starting inside its env: indices ceil(B/4) .. ceil((B+obe)/4) - 1
//   E[i] = bytes [q+4i, q+4i+4) of the env's stream continued by the next env's = funnel(D[i+1] : D[i], q)
// The one dword a lane cannot complete alone is its last: it ends with the first bytes of the next env, fetched from
// lane l+1's D[0] by the caller.  Host-callable so that the CPU suite can check it for every obe (mg_selftest_stream).
// ======================================================================================================
MG_HD uint32_t funnel_bytes(uint32_t hi, uint32_t lo, uint32_t q) {      // ({hi:lo} >> 8q), q in 0..3
#if defined(__HIP_DEVICE_COMPILE__)
  return __funnelshift_r(lo, hi, 8u * q);
#else
  return q ? (lo >> (8u * q)) | (hi << (32u - 8u * q)) : lo;
#endif
}
struct StreamEmit {
  uint32_t* p;        // next aligned dword of the group's stream this lane writes
  uint32_t prev;      // last dword handed in, not yet written out
  uint32_t q;
  uint32_t tb;        // valid bytes of the env's last dword D[ND-1] (1..4)
  // this lane's bytes are [B, B + len) of the stream (one env, or with several lanes per env its share of the env's cells)
  MG_HD void setup(uint32_t* stream, uint32_t B, uint32_t len) {
    q = (0u - B) & 3u;
    p = stream + ((B + q) >> 2);
    const uint32_t nd = (len + 3u) >> 2;
    tb = len - 4u * (nd - 1u);
  }
  MG_HD void first(uint32_t d0) { prev = d0; }
  MG_HD void put(uint32_t d) { *p++ = funnel_bytes(d, prev, q); prev = d; }
  // the env's last dword (tb valid bytes), completed with the first bytes of the next env's stream (next0 = its D[0])
  MG_HD void put_last(uint32_t d, uint32_t next0) {
    const uint32_t xl = tb < 4u ? ((d & ((1u << (8u * tb)) - 1u)) | (next0 << (8u * tb))) : d;
    const uint32_t xa = tb < 4u ? (next0 >> (32u - 8u * tb)) : next0;
    put(xl);
    if (q < tb) *p = funnel_bytes(xa, prev, q);          // one more dword starts inside this env
  }
};

// ======================================================================================================
// k_step: T lockstep steps of MiniGridEnv.step (minigrid_env.py:525-595) + RoomGridLevel.step/GoToInstr
// (roomgrid_level.py:87-104, verifier.py:309-316) + gen_obs (597-650: get_view_exts/slice/rotate_left/process_vis/
// encode) or FullyObsWrapper.observation (wrappers.py:419-426), with Gymnasium NEXT_STEP autoreset -- the loop of
// minigrid/benchmark.py:36-43 for 64 envs per wavefront.  T = 1 is Env.step(); T > 1 is the fused rollout.
// MODE 0 = partial VxVx3 view, 1 = FullyObs WxHx3, 2 = one-hot partial view VxVx20, 3 = symbolic WxHx3,
// 4 = tile map for k_render (RGBImgPartialObsWrapper: VxV bytes; RGBImgObsWrapper: WxH bytes), byte = tile key * 2 + highlight.
// (The default 7x7x3 view is k_roll7, mg_roll.h; MODE 0 here is the run-time view size of ViewSizeWrapper.)  GG = rule group compiled in (mg_gen.h).
//
// A wavefront is autonomous: lane l = env env0 + l, no workgroup barrier anywhere.  Launch start: the 64 grids are staged
// into LDS with 16 B/lane coalesced loads (and, for fused launches, each env's next spare episode into a shadow copy).
// Per step, all in registers + LDS: Philox action, dynamics (the one modified cell is written in place), view gather
// through a guard-banded layout (out-of-grid cells need no address clamp, only a mask), process_vis as bit-parallel rows,
// encode through a 256-entry LDS table, the wave's observations packed into aligned stream dwords (StreamEmit) and
// copied out with 16 B/lane stores, scalars with one coalesced store each.  HBM traffic per env-step in the loop: the
// outputs only (obe + 13 bytes written); the grid is read once and written back once per launch.
// ======================================================================================================
// LDS hand-off inside one wave for the step loop.  MG_WAVE_LDS_SYNC's release fence also waits for vmcnt(0) on gfx950 -- i.e.
// for every global store still in flight: inside the T-step loop that would serialise each step behind the previous step's
// observation stores (measured: 4.2 us per step instead of the store stream's own pace).  DS operations of one wave
// execute in order, so a compiler barrier plus an LDS-counter wait is all the hand-off needs.
#ifndef MG_EMU
#define MG_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define MG_LDS_SYNC() emu_wave_barrier()      // (tests/emu: the lanes of a wave run one after the other between cross-lane operations)
#endif

// The agent's view for any odd view size V <= 15 (ViewSizeWrapper; the default 7x7x3 view runs k_roll7, mg_roll.h), the one-hot
// encode (MODE 2) and the RGB tile map (MODE 4):
// run-time loops, visibility rows kept in LDS (16 x u16 per env), bytes stored straight at their stream position.
template <int MODE>
MG_D void obs_view_generic(const StepParams& P, const Agent& a, const uint8_t* mygrid, const uint32_t* slut, uint16_t* rows,
                           uint8_t* myT, bool active) {
  const int W = P.W, H = P.H;
  const int V = P.view, HV = V >> 1;
  const int fxv = dir_dx(a.dir), fyv = dir_dy(a.dir);
  const int rx = -fyv, ry = fxv;
  const bool horiz = fyv == 0;
  const uint32_t colmask = horiz ? inb_mask_v((int)a.y - HV * ry, ry, H, V) : inb_mask_v((int)a.x - HV * rx, rx, W, V);
  const uint32_t rowmask = horiz ? inb_mask_v((int)a.x + (V - 1) * fxv, -fxv, W, V) : inb_mask_v((int)a.y + (V - 1) * fyv, -fyv, H, V);
  const int SR = ry * W + rx, SU = -(fyv * W + fxv);
  const uint8_t* vbase = mygrid + ((int)a.y + (V - 1) * fyv - HV * ry) * W + ((int)a.x + (V - 1) * fxv - HV * rx);
  const uint32_t full = (1u << V) - 1u;
  auto cell_at = [&](int vx, int vy) -> uint32_t {
    const uint32_t raw = vbase[vy * SU + vx * SR];
    const uint32_t valid = 0u - (((rowmask >> vy) & (colmask >> vx)) & 1u);
    return ((raw ^ CELL_WALL_GREY) & valid) ^ CELL_WALL_GREY;
  };
  if (!P.see_through) {
#pragma unroll 1
    for (int vy = 0; vy < V; vy++) {
      uint32_t opq = 0;
      for (int vx = 0; vx < V; vx++) opq |= (cell_at(vx, vy) >> 7) << vx;
      rows[vy] = (uint16_t)(~opq & full);
    }
    uint32_t m = 1u << HV;
#pragma unroll 1
    for (int j = V - 1; j >= 0; j--) {
      uint32_t vr, up;
      vis_row_n(m, rows[j], V, &vr, &up);
      rows[j] = (uint16_t)vr;
      m = up;
    }
  }
#pragma unroll 1
  for (int vy = 0; vy < V; vy++) {
    const uint32_t vrow = P.see_through ? full : (uint32_t)rows[vy];
    for (int vx = 0; vx < V; vx++) {
      uint32_t c = cell_at(vx, vy);
      if (vx == HV && vy == V - 1) c = a.carry ? a.carry : (uint32_t)CELL_EMPTY;
      const uint32_t bit = (vrow >> vx) & 1u;
      if (MODE == 4) {
        // get_pov_render (minigrid_env.py:652-666): process_vis has blanked the invisible cells (grid.py:324-327),
        // so they are empty un-highlighted tiles (byte 0); visible ones are highlighted.  Image order [vy][vx].
        if (!P.rgb_full) myT[vy * V + vx] = (uint8_t)(slut[c] & (0u - bit));
        continue;
      }
      const uint32_t tri = slut[c & (0u - bit)];
      if (MODE == 0) {
        uint8_t* o = myT + (vx * V + vy) * 3;
        o[0] = (uint8_t)tri; o[1] = (uint8_t)(tri >> 8); o[2] = (uint8_t)(tri >> 16);
      } else if (active) {
        // OneHotPartialObsWrapper (wrappers.py:267-284): 20 bytes per cell, one 1 in each of the type / colour / state
        // groups.  Lanes past the batch end hold stale LDS "cells": their codes could index past the 20 bytes.
        uint8_t* o = myT + (vx * V + vy) * 20;
        uint32_t* o4 = (uint32_t*)o;
        o4[0] = 0; o4[1] = 0; o4[2] = 0; o4[3] = 0; o4[4] = 0;
        o[tri & 0xFF] = 1; o[11 + ((tri >> 8) & 0xFF)] = 1; o[17 + (tri >> 16)] = 1;
      }
    }
  }
  if (MODE == 4 && P.rgb_full) {
    // get_full_render (minigrid_env.py:668-714): every grid cell, highlighted where the agent's view sees it.
    // World cell (x, y) is view cell (HV + d.r, V-1 - d.f) with d = (x, y) - agent: the inverse of the gather above.
    const uint32_t hl_on = P.rgb_highlight ? 1u : 0u;
#pragma unroll 1
    for (int y = 0; y < H; y++) {
      const int dy = y - (int)a.y;
      for (int x = 0; x < W; x++) {
        const int idx = y * W + x, dx = x - (int)a.x;
        const uint32_t c = mygrid[idx];
        const int fwd = dx * fxv + dy * fyv, side = dx * rx + dy * ry + HV;
        const bool inside = (unsigned)fwd < (unsigned)V && (unsigned)side < (unsigned)V;
        const uint32_t vrow = P.see_through ? full : (uint32_t)rows[inside ? V - 1 - fwd : 0];
        const uint32_t bit = inside ? ((vrow >> side) & hl_on) : 0u;
        myT[idx] = (uint8_t)(slut[c] - 1u + bit);
      }
    }
  }
}

// MODE 1: FullyObsWrapper.observation (wrappers.py:419-426): grid.encode() in image[x][y] order, agent cell = (10, 0, dir).
// MODE 3: SymbolicObsWrapper.observation (wrappers.py:763-782): (x, y, type or -1), agent cell type = 10.
// Three bytes per cell in x-major order: 4 cells = 3 stream dwords, like the partial view.  LPE lanes per env: the
// cells / 4 units are dealt out in LPE consecutive runs (upl units each), the env's last lane also takes what is left over.
template <int MODE, int LPE>
MG_D void obs_full(const StepParams& P, const Agent& a, const uint8_t* mygrid, const uint32_t* slut, uint32_t* stream,
                   int lane, int nlanes) {
  const int W = P.W, H = P.H, cells = P.cells;
  const int sub = LPE == 1 ? 0 : (lane & (LPE - 1)), el = LPE == 1 ? lane : lane / LPE;
  const bool last_sub = sub == LPE - 1;
  const int units = cells >> 2, upl = units / LPE;               // host guarantees upl >= 1
  const int kc0 = sub * upl * 4;                                  // this lane's first cell (output order k = x * H + y)
  const int nunits = last_sub ? units - (LPE - 1) * upl : upl;
  const int rem = last_sub ? (cells & 3) : 0;
  const int aidx = (int)a.y * W + (int)a.x;
  int x = LPE == 1 ? 0 : kc0 / H, y = LPE == 1 ? 0 : kc0 - x * H;
  auto next_tri = [&]() -> uint32_t {
    const int idx = y * W + x;
    uint32_t c = mygrid[idx], r;
    if (MODE == 1) {
      if (idx == aidx) c = T_AGENT_MARK | (a.dir << 4);
      r = slut[c];
    } else {
      const uint32_t t = idx == aidx ? (uint32_t)T_AGENT : (c == CELL_EMPTY ? 0xFFu : cell_ref_type(c));
      r = (uint32_t)x | ((uint32_t)y << 8) | (t << 16);
    }
    if (++y == H) { y = 0; x++; }
    return r;
  };
  StreamEmit em;
  em.setup(stream, (uint32_t)(el * cells * 3 + kc0 * 3), (uint32_t)(nunits * 12 + rem * 3));
  uint32_t next0;
  {
    const uint32_t t0 = next_tri(), t1 = next_tri(), t2 = next_tri(), t3 = next_tri();
    const uint32_t d0 = t0 | (t1 << 24), d1 = (t1 >> 8) | (t2 << 16), d2 = (t2 >> 16) | (t3 << 8);
    em.first(d0);
    next0 = (uint32_t)__shfl_down((int)d0, 1);
    if (lane >= nlanes - 1) next0 = 0u;
    em.put(d1);
    if (nunits == 1 && rem == 0) em.put_last(d2, next0); else em.put(d2);
  }
#pragma unroll 1
  for (int u = 1; u < nunits; u++) {
    const uint32_t t0 = next_tri(), t1 = next_tri(), t2 = next_tri(), t3 = next_tri();
    const uint32_t d0 = t0 | (t1 << 24), d1 = (t1 >> 8) | (t2 << 16), d2 = (t2 >> 16) | (t3 << 8);
    em.put(d0); em.put(d1);
    if (u == nunits - 1 && rem == 0) em.put_last(d2, next0); else em.put(d2);
  }
  if (rem == 1) { em.put_last(next_tri(), next0); }
  else if (rem == 2) { const uint32_t t0 = next_tri(), t1 = next_tri(); em.put(t0 | (t1 << 24)); em.put_last(t1 >> 8, next0); }
  else if (rem == 3) { const uint32_t t0 = next_tri(), t1 = next_tri(), t2 = next_tri(); em.put(t0 | (t1 << 24)); em.put((t1 >> 8) | (t2 << 16)); em.put_last(t2 >> 16, next0); }
}

// ======================================================================================================
// One env transition: MiniGridEnv.reset (take the next spare episode) | observe-only | MiniGridEnv.step + the level rule,
// on the lane's registers (EnvRegs) and the env's LDS grid.  Shared by k_step and k_roll7 (mg_roll.h).
// ======================================================================================================
// LDS -> LDS copy of n dwords per lane, eight at a time: all eight reads are issued before the first write.  (Written as d[k] = s[k] the
// compiler must assume the two ranges overlap and waits for every read before the next write: one LDS round trip per dword -- 16 for an
// 8x8 grid, 0.85 us per reset of a GoToRedBall wave, whose 64 envs end an episode in nearly every step: profiles/r4/gotoredball_attr2.txt.)
MG_HD void lds_copy_dwords(uint32_t* d, const uint32_t* s, int n) {
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = s[k + i];
#pragma unroll
    for (int i = 0; i < 8; i++) d[k + i] = t[i];
  }
  if (k + 4 <= n) {
    uint32_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = s[k + i];
#pragma unroll
    for (int i = 0; i < 4; i++) d[k + i] = t[i];
    k += 4;
  }
  for (; k < n; k++) d[k] = s[k];
}

struct EnvRegs {          // what lives in registers across the steps of a launch
  Agent a;
  uint64_t targets, cur;  // GoTo levels: tracked positions / where the described objects are now (see below)
  uint32_t h;             // spares consumed so far (ring head)
  uint32_t shadow_left;   // staged spare episodes (LDS shadow slots) the env has not taken yet
  uint32_t ev_shadow;     // which shadow set the last reset took (ev_reset == 1)
  bool rec_dirty, aux_dirty, wb_all;
  uint32_t errbits;
  // what the last env_transition did to the env's grid (for kernels that keep a second image of it: k_roll7's FullyObs stream)
  int ev_dirty_idx;       // the one cell a step changed (row-major index), -1 = none
  uint32_t ev_dirty_code;
  uint32_t ev_reset;      // 0 = no reset, 1 = the staged shadow spare was taken, 2 = a spare was fetched from the ring in HBM
};
struct LaneCtx {          // the lane's view of its env: loop-invariant
  int e, el, sub;
  bool active, lead, reset_enabled, maskok, goto_rule;
  bool spare_in_lds = false; // k_roll7 (round 6): the wave has already copied this env's next spare grid from the ring into its LDS grid (cooperatively, 16 B per lane): take_spare skips its own copy
  bool spare_rec_pf = false; // ... and this lane has its next spare's agent record / auxiliary word in pf_agent / pf_aux (loaded one step ahead)
  uint64_t pf_agent = 0, pf_aux = 0;
  bool reset_only = false;   // only take the spare episode of the envs flagged RESET_PENDING (the sentence levels' SAME_STEP autoreset: their episodes end in the verifier, after the step)
  uint8_t* mygrid; const uint8_t* myshadow; const uint64_t* sspr;
};

// BabyAI GoTo levels (RoomGridLevel.step + GoToInstr): `targets` = the TRACKED positions of the described objects, which the
// reference refreshes from the grid on every drop ACTION (update_objs_poss, roomgrid_level.py:92-93) and never otherwise;
// `cur` = where the described objects are on the grid right now, kept up to date cell change by cell change, so that the
// refresh is `targets = cur` instead of a scan of the grid in every step in which some env of the wave drops.  The two
// differ only while a described object is carried; FLAG_TARGETS_STALE carries that fact across launches.
MG_HD uint32_t goto_desc(const StepParams& P, uint32_t mission) {
  // rule_div 0 = fixed cell code (rule_cell), 1 = red/blue ball by mission id, 2 = (colour, type) by mission id,
  // 3 = a door by colour (GoToDoor: id % 6), 4 = (colour, key | ball | box | door) (GoToObjDoor: id % 24).  A door description
  // is returned as the OPEN door's code and matches the door in any state (desc_match).  5 = ActionObjDoor: as 4; id / 48 = the verb.
  const uint32_t m18 = mission % 18u, m24 = mission % 24u;
  return P.rule_div == 0 ? (uint32_t)P.rule_cell
       : P.rule_div == 1 ? make_cell(T_BALL, mission ? (uint32_t)C_BLUE : (uint32_t)C_RED)
       : P.rule_div == 2 ? make_cell(T_KEY + m18 % 3u, color_from_sorted(m18 / 3u))
       : P.rule_div == 3 ? make_cell(T_DOOR, color_from_sorted(mission % 6u))
                         : make_cell((m24 & 3u) == 3u ? (uint32_t)T_DOOR : (uint32_t)T_KEY + (m24 & 3u), color_from_sorted(m24 >> 2));
}
MG_HD bool desc_match(uint32_t c, uint32_t desc) {
  return c == desc || (cell_type(desc) == T_DOOR && cell_ref_type(c) == T_DOOR && cell_color(c) == cell_color(desc));
}

// Host-callable (MG_HD): mg_selftest_transition runs the step core on the CPU against the oracle (tests/test_transition_cpu.py).
template <int GG, int LPE>
MG_HD void env_transition(const StepParams& P, const LaneCtx& C, EnvRegs& S, uint32_t act, double& reward, uint32_t& term, uint32_t& trunc) {
  const int W = P.W, H = P.H, CS = P.CS, cpe = P.CS >> 4;
  const size_t N = (size_t)P.N;
  const int e = C.e, sub = C.sub;
  const bool active = C.active, lead = C.lead, reset_enabled = C.reset_enabled, maskok = C.maskok, goto_rule = C.goto_rule;
  uint8_t* mygrid = C.mygrid;
  Agent& a = S.a;
  uint64_t& targets = S.targets; uint64_t& cur = S.cur;
  uint32_t& h = S.h; uint32_t& errbits = S.errbits;
  uint32_t& shadow_left = S.shadow_left; bool& rec_dirty = S.rec_dirty; bool& aux_dirty = S.aux_dirty; bool& wb_all = S.wb_all;
  // The one cell an action can change: it can only change under pickup/drop/toggle, which leave the pose alone, so it
  // is always the cell straight ahead.  The level rules below see the grid as it was BEFORE the action plus this patch
  // (RedBlueDoors compares both states); it is written into the LDS grid after them.
  int dirty_idx = -1;              // linear index of the modified cell, -1 = none
  uint32_t dirty_code = 0;
  S.ev_dirty_idx = -1; S.ev_dirty_code = 0; S.ev_reset = 0;
  // ---- MiniGridEnv.reset (minigrid_env.py:119-157): take the next spare episode out of the ring ----
  auto take_spare = [&]() {
      if (shadow_left == 0u) {
        // not staged (single-step launch, or the env took every staged spare of this fused launch already): straight from the ring in HBM into the
        // live LDS grid -- the loads and their wait stay inside this branch
        const size_t se = (size_t)(h & P.ring_mask) * N + (size_t)e;
        const uint4* src = (const uint4*)(P.spare_grid + se * CS);
        if (!C.spare_in_lds)
        for (int c = sub; c < cpe; c += LPE) {
          const uint4 v = src[c];
          uint32_t* d = (uint32_t*)(mygrid + c * 16);
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        a = agent_unpack(C.spare_rec_pf ? C.pf_agent : P.spare_agent[se]);
        if (goto_rule) { targets = C.spare_rec_pf ? C.pf_aux : P.spare_aux[se]; cur = targets; aux_dirty = true; }
        S.ev_reset = 2;
      } else {
        // shadow set 0 first, then set 1 (a level that draws nothing has ONE constant spare and takes it again and again)
        const uint32_t set = P.static_gen ? 0u : (uint32_t)P.use_shadow - shadow_left;
        const uint32_t* s = (const uint32_t*)(C.myshadow + set * (uint32_t)P.shadow_stride);
        const uint64_t* sp = (const uint64_t*)((const uint8_t*)C.sspr + set * (uint32_t)P.spr_stride);
        uint32_t* d = (uint32_t*)mygrid;
        if (LPE == 1) lds_copy_dwords(d, s, CS >> 2);
        else for (int k = sub; k < (CS >> 2); k += LPE) d[k] = s[k];
        a = agent_unpack(sp[0]);
        if (goto_rule) { targets = sp[1]; cur = targets; aux_dirty = true; }
        if (!P.static_gen) shadow_left--;
        S.ev_reset = 1; S.ev_shadow = set;
      }
      if (a.flags & FLAG_STUCK) errbits |= ERR_GENERATOR;   // the reference would never return from this reset() (mg_gen.h room_stuck)
      a.step = 0; a.flags &= FLAG_SHOW_TAKEN;           // (carrying: nothing, except PutNext's start_carrying episodes)
      if constexpr (gg_group(GG) == GG_NONE || gg_group(GG) == GG_SENTENCE) if (MG_RULE(GG, P) == RULE_SENTENCE) a.flags |= FLAG_NEW_EPISODE;   // k_verify / k_roll7<GG_SENTENCE> installs the instruction record
      rec_dirty = true; wb_all = true;
      if (!P.static_gen) h++;
  };
  if (active) {
    if ((a.flags & FLAG_RESET_PENDING) && reset_enabled && maskok && !MG_EXPBIT(P, 64)) {
      // (GG_DYNOBS has no spare ring: k_roll7 redraws the env in place before it gets here, the host's live refill before an observe launch)
      if constexpr (gg_group(GG) == GG_DYNOBS) errbits |= ERR_GENERATOR; else take_spare();
    } else if (C.reset_only) {
    } else if (a.flags & FLAG_FRESH) {
      a.flags &= ~(FLAG_FRESH | FLAG_NOT_CLEAR);       // drawn by the generator launch just before this one: observe only
      rec_dirty = true;
    } else if (P.phase == PHASE_STEP) {
      // ---- MiniGridEnv.step (minigrid_env.py:525-595) ----
      // STRAIGHT-LINE: under a random policy every action occurs among a wave's 64 envs in every step, so an if / else chain over the
      // action makes the wave walk all seven bodies one after the other (exec-mask save / restore + branch around each: the round-3
      // transition cost 1.1 of the 2.7 us of a 65 536-env step, profiles/r4/attribution.txt, more than the observation).  Every effect
      // is computed for every lane and selected by the action instead; the only memory access is the front cell.
      rec_dirty = true;
      const uint32_t pre_carry = a.carry;
      a.step = a.step + 1u < 0xFFFFu ? a.step + 1u : 0xFFFFu;
      const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
      const bool inb = (unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H;
      if (!inb) errbits |= ERR_OOB;                                  // reference asserts (core/grid.py:74-78)
      const uint32_t fidx = inb ? (uint32_t)(fy * W + fx) : 0u;
      const uint32_t Fraw = (uint32_t)mygrid[fidx];
      const uint32_t F = inb ? Fraw : (uint32_t)CELL_WALL_GREY;
      const uint32_t ftype = cell_type(F), fcol = F & 0x70u;
      const bool is_fwd = act == A_FORWARD, is_pick = act == A_PICKUP, is_drop = act == A_DROP, is_tog = act == A_TOGGLE;
      if (act > A_DONE) errbits |= ERR_BAD_ACTION;                   // reference raises ValueError (584-585)
      // left / right (:552-560)
      a.dir = (a.dir + (act == A_LEFT ? 3u : act == A_RIGHT ? 1u : 0u)) & 3u;
      // forward (:563-571)
      const bool walk = is_fwd && cell_walkable(F);
      a.x = walk ? (uint32_t)fx : a.x; a.y = walk ? (uint32_t)fy : a.y;
      bool success = is_fwd && ftype == T_GOAL;
      if (is_fwd && (ftype == T_GOAL || ftype == T_LAVA)) term = 1;
      // pickup (:574-579) / drop (:582-587)
      const bool picks = is_pick && cell_pickable(F) && pre_carry == 0u;
      const bool drops = is_drop && F == CELL_EMPTY && pre_carry != 0u;
      // toggle (:590-592): Door.toggle / Box.toggle (world_object.py:184-194, 290-293) as a table over the type nibble --
      // open door (4) -> closed (11), closed -> open, locked (12) -> open with the key of its colour in hand, box (7) -> gone,
      // box-with-key (14) -> the key (5) -- everything else toggles to itself
      constexpr uint64_t TOG = 0xF5D44A98165B3210ull;   // nibble t = the type after a toggle
      static_assert(((TOG >> (4 * T_DOOR)) & 15) == T_DOOR_CLOSED && ((TOG >> (4 * T_DOOR_CLOSED)) & 15) == T_DOOR && ((TOG >> (4 * T_BOX)) & 15) == T_EMPTY &&
                    ((TOG >> (4 * T_BOX_KEY)) & 15) == T_KEY && ((TOG >> (4 * T_DOOR_LOCKED)) & 15) == T_DOOR && ((TOG >> (4 * T_WALL)) & 15) == T_WALL &&
                    ((TOG >> (4 * T_BOX_DOORKEY)) & 15) == T_BOX_DOORKEY && ((TOG >> (4 * T_KEY)) & 15) == T_KEY, "toggle table");
      const bool no_key = ftype == T_DOOR_LOCKED && pre_carry != ((uint32_t)T_KEY | fcol);
      const uint32_t tt = no_key ? (uint32_t)T_DOOR_LOCKED : (uint32_t)(TOG >> (4u * ftype)) & 15u;
      const uint32_t toggled = tt | (tt == T_EMPTY ? 0u : fcol) | (((OPAQUE_TYPES >> tt) & 1u) << 7);
      uint32_t newF = is_tog ? toggled : picks ? (uint32_t)CELL_EMPTY : drops ? pre_carry : F;
      a.carry = picks ? F : drops ? 0u : pre_carry;
      if constexpr (gg_group(GG) == GG_ROOMS) if (is_tog && ftype == T_BOX_DOORKEY) {
        // KeyInBox: Box.toggle leaves what the box contains, the key of the level's only door (world_object.py:290-293)
        uint32_t dc = 0;
        for (int k = 0; k < P.cells; k++) { const uint32_t c = mygrid[k]; if (cell_ref_type(c) == T_DOOR) dc = cell_color(c); }
        newF = make_cell(T_KEY, dc);
      }
      if (newF != F && inb) { dirty_idx = (int)fidx; dirty_code = newF; }
      trunc = a.step >= (uint32_t)P.max_steps;
      if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_GOTO && !MG_EXPBIT(P, 128)) {
        // RoomGridLevel.step (roomgrid_level.py:87-104) + GoToInstr.verify_action (verifier.py:309-316): success iff
        // the post-action front cell is one of the TRACKED POSITIONS of the described objects.  They are positions,
        // not objects: refreshed only at reset and after a drop (update_objs_poss), so they go stale while a target is
        // carried -- which only matters when a finished episode keeps being stepped (autoreset disabled).
        if (dirty_idx >= 0) {
          const uint32_t desc = goto_desc(P, a.mission);
          if (F == desc) cur &= ~(1ull << dirty_idx);
          if (newF == desc) cur |= 1ull << dirty_idx;
        }
        if (act == A_DROP && targets != cur) { targets = cur; aux_dirty = true; }
        const int gx = (int)a.x + dir_dx(a.dir), gy = (int)a.y + dir_dy(a.dir);
        if ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && ((targets >> (gy * W + gx)) & 1ull)) { term = 1; success = true; }
      }
      if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_GOTO_BIG) {
        // GoToInstr on grids of more than 64 cells (the multi-room BabyAI GoTo levels).  Tracked positions T = the cells holding a
        // described object at the last refresh (reset, every drop ACTION).  Between refreshes nothing can add such a cell (only
        // a drop does, and a drop refreshes), so T = {cells holding the object NOW} + S, S = where one was removed since (picked up,
        // or a box toggled away): at most one pickup plus the toggled boxes.  `targets` = S as four 16-bit cell indices
        // (0xFFFF = free); a fifth is reported as ERR_TRACKED instead of being dropped silently.
        const uint32_t desc = goto_desc(P, a.mission);
        if (dirty_idx >= 0 && desc_match(F, desc) && !desc_match(newF, desc)) {
          int slot = -1;
#pragma unroll
          for (int k = 3; k >= 0; k--) if (((targets >> (16 * k)) & 0xFFFFull) == 0xFFFFull) slot = k;
          if (slot < 0) errbits |= ERR_TRACKED;
          else targets = (targets & ~(0xFFFFull << (16 * slot))) | ((uint64_t)dirty_idx << (16 * slot));
          aux_dirty = true;
        }
        if (act == A_DROP && targets != ~0ull) { targets = ~0ull; aux_dirty = true; }           // update_objs_poss
        const int gx = (int)a.x + dir_dx(a.dir), gy = (int)a.y + dir_dy(a.dir);
        if ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H) {
          const int gi = gy * W + gx;
          const uint32_t c = gi == dirty_idx ? dirty_code : (uint32_t)mygrid[gi];
          bool hit = desc_match(c, desc);
#pragma unroll
          for (int k = 0; k < 4; k++) hit |= ((targets >> (16 * k)) & 0xFFFFull) == (uint64_t)gi;
          if (hit && (P.rule_div != 5 || a.mission < 48u)) { term = 1; success = true; }
        }
        if (P.rule_div == 5 && a.mission >= 48u) {
          // ActionObjDoor (other.py:86-106): PickupInstr / OpenInstr about the same description (verifier.py:343-363, 270-287)
          if (a.mission < 96u) { if (act == A_PICKUP && pre_carry == 0 && a.carry == desc) { term = 1; success = true; } }
          else if (act == A_TOGGLE && inb && cell_type(newF) == T_DOOR && cell_color(newF) == cell_color(desc)) { term = 1; success = true; }
        }
      }
      if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_PUTNEXT && act == A_DROP && pre_carry != 0 && newF != F) {
        // RoomGridLevel.step + PutNextInstr.verify_action (verifier.py:406-431): this drop put down the object to move
        // (preCarrying is it; every object of these levels is the only one of its type and colour) and it now lies next to
        // (Manhattan distance 1) the fixed object, wherever that is NOW (update_objs_poss runs on every drop action).  A drop
        // that fails leaves cur_pos at (-1, -1) or -- start_carrying -- at the initial cell, which validate_instrs made non-adjacent
        // to the fixed object, and that object cannot have moved while the hands were full.
        // start_carrying (rule_div == 1): the verifier was reset before the object was handed over, so at the episode's first
        // step its preCarrying is still None
        const uint32_t mv = a.mission / 18u, fo = a.mission % 18u;
        if (!(P.rule_div == 1 && a.step == 1u) && pre_carry == make_cell((uint32_t)T_KEY + mv % 3u, color_from_sorted(mv / 3u))) {
          const uint32_t fixed = make_cell((uint32_t)T_KEY + fo % 3u, color_from_sorted(fo / 3u));
          bool next = false;
#pragma unroll 1
          for (int d = 0; d < 4; d++) {
            const int nx = fx + dir_dx((uint32_t)d), ny = fy + dir_dy((uint32_t)d);
            if ((unsigned)nx < (unsigned)W && (unsigned)ny < (unsigned)H) next |= (uint32_t)mygrid[ny * W + nx] == fixed;
          }
          if (next) { term = 1; success = true; }
        }
      }
      if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_OPENDOOR && act == A_TOGGLE && inb) {
        // OpenInstr.verify_action incl. strict mode (verifier.py:270-287): the described doors = `targets`, a COLOR_TO_IDX bit mask
        // fixed at reset (by colour, or by where the doors were relative to the agent then; the four doors' colours differ)
        if (cell_type(newF) == T_DOOR && ((targets >> cell_color(newF)) & 1ull)) { term = 1; success = true; }
        else if (P.rule_div == 1 && cell_ref_type(newF) == T_DOOR) term = 1;
      }
      if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_GOTOOBJ) {
        // GoToObjectEnv.step (gotoobject.py:137-153): toggle ends the episode; done ends it, rewarded when the agent
        // stands next to target_pos (the one-bit board drawn at reset)
        if (act == A_TOGGLE) term = 1;
        if (act == A_DONE) {
          const int ax = (int)a.x, ay = (int)a.y;          // interior cell: the four neighbours are inside the grid
          const uint64_t ring = (1ull << (ay * W + ax - 1)) | (1ull << (ay * W + ax + 1)) | (1ull << ((ay - 1) * W + ax)) | (1ull << ((ay + 1) * W + ax));
          term = 1; success = (targets & ring) != 0;
        }
      }
      if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_PUTNEAR) {
        // PutNearEnv.step (putnear.py:177-199).  Mission id = ((move colour * 3 + move type) * 6 + target colour) * 3 + target type
        // (COLOR_NAMES / [key, ball, box] indices); target_pos is a POSITION fixed at reset: the one-bit board `targets`.
        const uint32_t mv = a.mission / 18u;
        const uint32_t move = make_cell((uint32_t)T_KEY + mv % 3u, color_from_sorted(mv / 3u));
        if (act == A_PICKUP && a.carry != 0 && a.carry != move) term = 1;           // picked up the wrong object
        if (act == A_DROP && pre_carry != 0) {
          if (newF != F && inb && targets) {                                        // `grid.get(ox, oy) is preCarrying`: the drop happened
            const int tidx = __builtin_ffsll((long long)targets) - 1, tx = tidx % W, ty = tidx / W;
            if (abs(fx - tx) <= 1 && abs(fy - ty) <= 1) success = true;
          }
          term = 1;
        }
      }
      if constexpr (gg_group(GG) == GG_LIGHT) if (MG_RULE(GG, P) == RULE_FETCH && a.carry != 0) {
        // FetchEnv.step (fetch.py:162-175): carrying anything ends the episode; the target (type, colour) is encoded
        // in the mission id = syntax*12 + COLOR_NAMES index*2 + (key 0 | ball 1)
        const uint32_t m12 = a.mission % 12u;
        const uint32_t target = make_cell((m12 & 1u) ? (uint32_t)T_BALL : (uint32_t)T_KEY, color_from_sorted(m12 >> 1));
        term = 1; success = a.carry == target;
      }
      if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_UNLOCK && act == A_TOGGLE) {
        // UnlockEnv.step (unlock.py:90-98): after a toggle, success iff THE door is open.  The level has one door, in
        // the wall column between the two rooms (x = rule_cell); scanning the column is exact even past termination.
        bool open = false;
        for (int y = 1; y < H - 1; y++) {
          const int idx = y * W + P.rule_cell;
          const uint32_t c = idx == dirty_idx ? dirty_code : (uint32_t)mygrid[idx];
          open |= cell_type(c) == T_DOOR;
        }
        if (open) { term = 1; success = true; }
      }
      if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_PICKUP && act == A_PICKUP && a.carry != 0) {
        // UnlockPickupEnv.step (unlockpickup.py:99-107) & co.: `self.carrying == self.obj`; the target is the only
        // object of its (type, colour): type = rule_cell, colour from the mission id / rule_div
        const uint32_t target = make_cell((uint32_t)P.rule_cell, color_from_sorted(a.mission / (uint32_t)P.rule_div));
        if (a.carry == target) { term = 1; success = true; }
      }
      if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_PICKUPDESC && act == A_PICKUP && a.carry != 0) {
        // RoomGridLevel.step + PickupInstr.verify_action (roomgrid_level.py:87-104, verifier.py:343-363): success iff the
        // object was picked up by THIS action (preCarrying is None) and matches the description the mission id encodes
        // (desc.obj_set = the objects matching at reset; attributes never change, so membership = matching);
        // strict (PickupDistDebug, rule_div == 2): any other pickup action with something in hand fails the episode
        const uint32_t m = a.mission % 28u, ci = m >> 2, ti = m & 3u;
        const bool match = (ti == 0u || cell_type(a.carry) == (uint32_t)T_KEY + ti - 1u) &&
                           (ci == 0u || cell_color(a.carry) == color_from_sorted(ci - 1u));
        if (newF != F && match) { term = 1; success = true; }
        else if (P.rule_div == 2) term = 1;
      }
      if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_OPENFRONT && act == A_TOGGLE) {
        // OpenInstr.verify_action (verifier.py:270-287): the cell in front is the described door (the level's only one)
        // and it is open after the toggle
        // (rule_div == 6: the description names a colour -- mission id % 6 -- and any door of that colour counts)
        if (inb && cell_type(newF) == T_DOOR && (P.rule_div != 6 || cell_color(newF) == color_from_sorted(a.mission % 6u))) { term = 1; success = true; }
      }
      if constexpr (gg_group(GG) == GG_LIGHT) if (MG_RULE(GG, P) == RULE_REDBLUE) {
        // RedBlueDoorsEnv.step (redbluedoors.py:104-126): open states of the two doors before / after the action.
        // The doors sit somewhere in the two inner wall columns (x = H/2 and H/2 + H - 1).
        bool red_before = false, red_after = false, blue_before = false, blue_after = false;
        const int xr = H / 2, xb = H / 2 + H - 1;
#pragma unroll 1
        for (int y = 1; y < H - 1; y++) {
          const int ir = y * W + xr, ib = y * W + xb;
          const uint32_t r0 = mygrid[ir], b0 = mygrid[ib];
          const uint32_t r1 = ir == dirty_idx ? dirty_code : r0, b1 = ib == dirty_idx ? dirty_code : b0;
          red_before |= cell_type(r0) == T_DOOR; red_after |= cell_type(r1) == T_DOOR;
          blue_before |= cell_type(b0) == T_DOOR; blue_after |= cell_type(b1) == T_DOOR;
        }
        if (blue_after) { term = 1; success = red_before; }
        else if (red_after && blue_before) { term = 1; success = false; }
      }
      if constexpr (gg_group(GG) == GG_LIGHT) if (MG_RULE(GG, P) == RULE_MEMORY) {
        // MemoryEnv.step (memory.py:155-162): success_pos / failure_pos are the two hallway-end cells next to the
        // objects at (hallway_end + 1, H/2 -+ 2); nothing can move those objects (pickup is remapped to toggle), so
        // "the agent stands at H/2 -+ 1 right below/above a key or ball" identifies them, and the match is decided by
        // the start-room object at (1, H/2 - 1)
        const int mid = H / 2;
        const int oy = (int)a.y == mid - 1 ? mid - 2 : ((int)a.y == mid + 1 ? mid + 2 : -1);
        if (oy >= 0) {
          const uint32_t o = mygrid[oy * W + (int)a.x], st = mygrid[(mid - 1) * W + 1];
          if (cell_type(o) == T_KEY || cell_type(o) == T_BALL) { term = 1; success = cell_type(o) == cell_type(st); }
        }
      }
      if constexpr (gg_group(GG) == GG_LIGHT) if (MG_RULE(GG, P) == RULE_GOTODOOR) {
        // GoToDoorEnv.step (gotodoor.py:133-149): toggle ends the episode; done ends it, rewarded next to the target
        // door = the door whose colour the mission names (door colours are distinct and doors never move)
        if (act == A_TOGGLE) term = 1;
        if (act == A_DONE) {
          const uint32_t tc = color_from_sorted(a.mission);
          bool next_to = false;
#pragma unroll 1
          for (int d = 0; d < 4; d++) {
            const int nx = (int)a.x + dir_dx((uint32_t)d), ny = (int)a.y + dir_dy((uint32_t)d);
            if ((unsigned)nx < (unsigned)W && (unsigned)ny < (unsigned)H) {
              const uint32_t c = mygrid[ny * W + nx];
              next_to |= cell_ref_type(c) == T_DOOR && cell_color(c) == tc;
            }
          }
          term = 1; success = next_to;
        }
      }
      if (P.done_actions && MG_RULE(GG, P) != RULE_SENTENCE) {            // (the sentence levels: inside verify_action, per leaf)
        // ActionInstr.verify with use_done_actions (verifier.py:228-242), levels with ONE action instruction (the rules above are its
        // verify_action): `done` reports success iff the previous action completed the instruction, else failure; every other action only
        // remembers whether it matched (the method returns None: RoomGridLevel.step carries on).  The host sets the switch for the
        // RoomGridLevel-based levels only.
        const bool matched = term != 0u && success;
        if (act == A_DONE) { term = 1; success = (a.flags & FLAG_LAST_MATCH) != 0u; }
        else { a.flags = matched ? (a.flags | FLAG_LAST_MATCH) : (a.flags & ~FLAG_LAST_MATCH); term = 0; success = false; }
      }
      if (success) reward = reward_exact(a.step, P.max_steps);       // three IEEE-rounded f64 operations, like CPython's
      if constexpr (gg_group(GG) == GG_NONE || gg_group(GG) == GG_DYNOBS) if (gg_group(GG) == GG_DYNOBS || MG_RULE(GG, P) == RULE_DYNOBS) {
        // DynamicObstaclesEnv.step (dynamicobstacles.py:162-165): walking into what WAS an obstacle or wall before the
        // obstacles moved (k_move_obstacles / k_roll7<GG_DYNOBS> recorded it) costs -1 and ends the episode, whatever happened since
        if (act == A_FORWARD && (a.flags & FLAG_NOT_CLEAR)) { reward = -1.0; term = 1; }
        a.flags &= ~FLAG_NOT_CLEAR;
      }
      if (P.no_death_mask && term) {
        // NoDeath.step (wrappers.py:860-882): walking into (or ending the episode while standing in) a no-death
        // cell does not terminate; death_cost is added to the reward instead
        const bool going = act == A_FORWARD && F != CELL_EMPTY && ((P.no_death_mask >> cell_ref_type(F)) & 1);
        const uint32_t U = mygrid[(int)a.y * W + (int)a.x];
        const bool in_death = U != CELL_EMPTY && ((P.no_death_mask >> cell_ref_type(U)) & 1);
        if (going || in_death) { term = 0; reward = dadd_rn(reward, P.death_cost); }
      }
      if ((term | trunc) && P.autoreset_next_step) a.flags |= FLAG_RESET_PENDING;
      if (dirty_idx >= 0) {
        S.ev_dirty_idx = dirty_idx; S.ev_dirty_code = dirty_code;
        if (lead) mygrid[dirty_idx] = (uint8_t)dirty_code;
        if (P.T == 1) { if (lead) P.grid[(size_t)e * CS + dirty_idx] = (uint8_t)dirty_code; }
        else wb_all = true;
      }
      // Gymnasium's SAME_STEP autoreset (the vector semantics of gymnasium 0.28 / 0.29, which the reference pins as its minimum): the step
      // that ends an episode also resets the env; the observation returned is the new episode's first, reward / terminated / truncated
      // are the ended episode's
      // (GG_DYNOBS: k_roll7 redraws the env in place right after this call)
      if constexpr (gg_group(GG) != GG_DYNOBS) if ((term | trunc) && P.autoreset_same_step) { S.ev_dirty_idx = -1; take_spare(); }
    }
  }
}

template <int MODE, int GG, int LPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))    // one autonomous wave per workgroup; LDS, not registers, bounds the occupancy
k_step(const StepParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // LPE lanes per env (1 or 4): lane = el * LPE + sub.  The LPE lanes of an env hold the same agent state and compute the
  // same dynamics (free in SIMD terms); they share the observation work (cells dealt out in stream order) and sub-lane 0
  // does the env's stores.  Fewer envs per wave = more waves for the same batch: latency hiding without any barrier.
  constexpr int EPW = 64 / LPE;
  const int lane = threadIdx.x;
  const int el = LPE == 1 ? lane : lane / LPE, sub = LPE == 1 ? 0 : (lane & (LPE - 1));
  const bool lead = sub == 0;
  const int wg = blockIdx.x;
  const int env0 = wg * EPW;
  const int e = env0 + el;
  const bool active = e < P.N;
  const int nvalid = min(EPW, P.N - env0);
  const int CS = P.CS, GS = P.GS;
  const size_t N = (size_t)P.N;
  uint32_t* slut = (uint32_t*)smem;                              // 256-entry cell code -> (type, colour, state) / tile key table
  uint8_t* sgrid = smem + P.off_grid;
  uint8_t* sshadow = smem + P.off_shadow;
  uint16_t* srows = (uint16_t*)(smem + P.off_trow) + el * 16;
  uint8_t* sslot = smem + P.off_trow;                            // staging only: ring slot of each env's next spare
  uint64_t* sspr = (uint64_t*)(smem + P.off_spr) + el * 2;       // shadow slot: the spare's agent record and auxiliary word
  uint8_t* sact = smem + P.off_act;                              // caller-supplied actions of the launch's steps: [T][EPW]
  uint8_t* sT = smem + P.off_T;
  const bool reset_enabled = P.autoreset_next_step || P.phase == PHASE_OBSERVE;
  // levels with an auxiliary word.  The single-room rules share the variant of the BASELINE GoToRedBall config; the heavier multi-room
  // ones live in the GG_ROOMS variants so that they do not cost it registers (202 VGPRs with everything in one variant)
  const bool goto_rule = (gg_group(GG) == GG_ROOMGRID && (MG_RULE(GG, P) == RULE_GOTO || MG_RULE(GG, P) == RULE_GOTOOBJ || MG_RULE(GG, P) == RULE_PUTNEAR)) ||
                         (gg_group(GG) == GG_ROOMS && (MG_RULE(GG, P) == RULE_GOTO_BIG || MG_RULE(GG, P) == RULE_PUTNEXT || MG_RULE(GG, P) == RULE_OPENDOOR));

  // ---- every independent load is issued up front ----
  const uint64_t rec = active ? P.agent[e] : 0ull;
  EnvRegs S;
  uint64_t& targets = S.targets; uint64_t& cur = S.cur; uint32_t& h = S.h; uint32_t& errbits = S.errbits;
  uint32_t& shadow_left = S.shadow_left; bool& rec_dirty = S.rec_dirty; bool& aux_dirty = S.aux_dirty; bool& wb_all = S.wb_all;
  Agent& a = S.a;
  targets = (goto_rule && active) ? P.aux[e] : 0ull;   // BabyAI GoTo levels: tracked positions
  h = (P.head && active) ? P.head[e] : 0u;
  const uint32_t h_in = h;
  uint32_t qn = P.seg_count ? uni32(P.seg_count[wg]) : 0u;
  const bool maskok = !P.obs_mask || (active && P.obs_mask[e]);
  // No global LOAD may sit in the step loop or feed a value that is live across it: on gfx950 loads and stores share vmcnt,
  // so the s_waitcnt a (even conditional, even never-taken) load needs at its join point waits for every observation store
  // still in flight -- one HBM round trip per step.  Everything the loop may read is staged in LDS here: the grids, the
  // next spare episode (shadow slot), the caller's actions; the success reward is computed, not looked up.
  shadow_left = P.use_shadow != 0 ? 1u : 0u;
  if (shadow_left && active && lead) {
    const size_t se = (size_t)(h & P.ring_mask) * N + (size_t)e;
    sspr[0] = P.spare_agent[se];
    sspr[1] = goto_rule ? P.spare_aux[se] : 0ull;
  }
  if (P.phase == PHASE_STEP && P.act_src == ACT_SRC_BUFFER && active && lead)
    for (int j = 0; j < P.T; j++) sact[j * EPW + el] = (uint8_t)load_action(P, e, j);
#pragma unroll
  for (int k = lane; k < 256; k += 64) slut[k] = MODE == 4 ? cell_tile_key((uint32_t)k) * 2u + 1u : cell_triple((uint32_t)k);
  const int cpe = CS >> 4;
  const int nchunks = nvalid * cpe;
  if (P.use_shadow) { if (lead) sslot[el] = (uint8_t)(h & P.ring_mask); MG_LDS_SYNC(); }
  {
    // stage the 64 grids: 16 B per lane, fully coalesced (and the next spare episode of every env next to it)
    const uint4* live = (const uint4*)(P.grid + (size_t)env0 * CS);
    for (int c = lane; c < nchunks; c += 64) {
      const uint32_t ce = ((uint32_t)c * P.cpe_magic) >> 20;
      const uint32_t part = (uint32_t)c - ce * (uint32_t)cpe;
      const uint4 v = live[c];
      uint32_t* dst = (uint32_t*)(sgrid + ce * GS + part * 16);
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      if (P.use_shadow) {
        const uint32_t slot = sslot[ce];         // (not a __shfl: lanes past the last chunk are off here, and read as 0)
        const uint4 s = ((const uint4*)(P.spare_grid + ((size_t)slot * N + (size_t)env0 + ce) * CS))[part];
        uint32_t* d2 = (uint32_t*)(sshadow + ce * GS + part * 16);
        d2[0] = s.x; d2[1] = s.y; d2[2] = s.z; d2[3] = s.w;
      }
    }
  }
  MG_LDS_SYNC();

  a = agent_unpack(rec);
  uint8_t* mygrid = sgrid + el * GS;
  cur = targets;
  if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_GOTO && (a.flags & FLAG_TARGETS_STALE)) {
    const uint32_t desc = goto_desc(P, a.mission);
    cur = 0;
    for (int k = 0; k < P.cells; k++) cur |= (uint64_t)((uint32_t)mygrid[k] == desc) << k;
  }
  // per-lane byte offset of this env's mg_step_scalars inside a trajectory slot (slot_bytes < 4 GB): a vector register
  const uint32_t o_scal = (uint32_t)P.off_reward + (uint32_t)e * 16u;
  rec_dirty = false; aux_dirty = false; wb_all = false; errbits = 0;
  uint32_t fin_total = 0;
  LaneCtx C;
  C.e = e; C.el = el; C.sub = sub; C.active = active; C.lead = lead; C.reset_enabled = reset_enabled; C.maskok = maskok; C.goto_rule = goto_rule;
  C.mygrid = mygrid; C.myshadow = sshadow + el * GS; C.sspr = sspr;
  uint32_t pw[4] = { 0, 0, 0, 0 };

  for (int j = 0; j < P.T; j++) {
    int slot_out = P.slot0 - j;
    slot_out += slot_out < 0 ? P.S : 0;                              // (T <= S, slot0 < S: at most one wrap)
    uint8_t* ob = P.out + (size_t)slot_out * P.slot_bytes;
    // ---- action ----
    uint32_t act = A_DONE;
    if (P.phase == PHASE_STEP) {
      if (P.act_src == ACT_SRC_PHILOX) {
        const uint32_t t = P.t0 + (uint32_t)j;
        if (j == 0 || (t & 3u) == 0u) philox_action_block(P, e, t >> 2, pw);
        const uint32_t w = (t & 3u) == 0u ? pw[0] : (t & 3u) == 1u ? pw[1] : (t & 3u) == 2u ? pw[2] : pw[3];
        act = (uint32_t)(((uint64_t)w * 7u) >> 32);
      } else act = sact[j * EPW + el];
    }
    const uint32_t act_in = act;
    if constexpr (gg_group(GG) == GG_LIGHT) if (MG_RULE(GG, P) == RULE_MEMORY && act == A_PICKUP) act = A_TOGGLE;    // MemoryEnv.step (memory.py:151-153)
    if constexpr (gg_group(GG) == GG_NONE) if (MG_RULE(GG, P) == RULE_DYNOBS && act >= 3u) act = A_LEFT;             // "Invalid action" (dynamicobstacles.py:137-139)
    double reward = 0.0;
    uint32_t term = 0, trunc = 0;
    env_transition<GG, LPE>(P, C, S, act, reward, term, trunc);
    if (P.phase == PHASE_STEP) fin_total += (uint32_t)__popcll(__ballot(active && lead && (term | trunc)));   // episodes finished in this wave
    MG_LDS_SYNC();

    // ---- per-env scalar outputs: one coalesced store each ----
    if (active && lead) {
      uint4 v;                                                       // mg_step_scalars (include/minigrid_hip.h): one 16-byte store
      v.x = (uint32_t)__double2loint(reward); v.y = (uint32_t)__double2hiint(reward);
      v.z = term | (trunc << 8) | (a.dir << 16) | (act_in << 24);
      v.w = a.mission & 0xFFFFu;
      *(uint4*)(ob + o_scal) = v;
    }

    // ---- observation -> the wave's byte stream in LDS ----
    const int obe = P.OBE;
    Agent av = a;
    bool show_taken = false;
    if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_PUTNEXT && active && (a.flags & FLAG_SHOW_TAKEN)) {
      // PutNext(start_carrying).reset (putnext.py:205-214) takes the object off the grid AFTER MiniGridEnv.reset made the
      // observation: the episode's first core observation (and what OneHotPartialObsWrapper makes of it) shows it where it was, and
      // empty hands.  The wrappers that look at the env when they are called (FullyObs, Symbolic, RGBImg*) see the state after.
      if constexpr (MODE == 0 || MODE == 2) {
        show_taken = true; av.carry = 0;
        if (lead) mygrid[(int)(targets & 0xFFFFull)] = (uint8_t)a.carry;
      }
      a.flags &= ~FLAG_SHOW_TAKEN; rec_dirty = true;
    }
    if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_PUTNEXT) MG_LDS_SYNC();
    if constexpr (MODE == 0 || MODE == 2 || MODE == 4) {
      static_assert(MODE == 1 || MODE == 3 || LPE == 1, "the generic view encode runs one lane per env");
      obs_view_generic<MODE>(P, av, mygrid, slut, srows, sT + el * obe, active);
    } else {
      obs_full<MODE, LPE>(P, av, mygrid, slut, (uint32_t*)sT, lane, nvalid * LPE);
    }
    MG_LDS_SYNC();
    if constexpr (gg_group(GG) == GG_ROOMS) if (show_taken && lead) mygrid[(int)(targets & 0xFFFFull)] = (uint8_t)CELL_EMPTY;

    // ---- the wave's observations are one contiguous byte stream in LDS and in HBM: 16 B per lane per store ----
    {
      uint8_t* obase = P.obs + (size_t)slot_out * P.obs_stride + (size_t)env0 * (size_t)obe;    // 64*obe is a multiple of 16
      const int nbytes = nvalid * obe;
      const int nvec = nbytes >> 4;
      {
#pragma unroll 4
        for (int c = lane; c < nvec; c += 64) ((uint4*)obase)[c] = ((const uint4*)sT)[c];
        for (int b = (nvec << 4) + lane; b < nbytes; b += 64) obase[b] = sT[b];   // ragged last group only
      }
    }
    MG_LDS_SYNC();
  }

  // ---- launch end: state back to HBM, refill requests, statistics ----
  if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_GOTO) {
    const uint32_t fl = (a.flags & ~FLAG_TARGETS_STALE) | (cur != targets ? FLAG_TARGETS_STALE : 0u);
    if (fl != a.flags) { a.flags = fl; rec_dirty = true; }
  }
  if (active && lead) {
    if (rec_dirty) P.agent[e] = agent_pack(a);
    if (goto_rule && aux_dirty) P.aux[e] = targets;
    // (sentence levels: k_verify publishes head, after it copied the consumed slot's instruction record -- a refill running on the
    // generator stream treats every slot below head + R as free the moment head moves)
    if (h != h_in && !(gg_group(GG) == GG_NONE && MG_RULE(GG, P) == RULE_SENTENCE)) P.head[e] = h;
    if (errbits) report_errors(P.err, errbits);
  }
  {
    const unsigned long long wb = __ballot(active && lead && wb_all);      // envs whose whole live grid changed (new episode, fused launch)
    if (wb) {
      uint4* live = (uint4*)(P.grid + (size_t)env0 * CS);
      for (int c = lane; c < nchunks; c += 64) {
        const uint32_t ce = ((uint32_t)c * P.cpe_magic) >> 20;
        const uint32_t part = (uint32_t)c - ce * (uint32_t)cpe;
        if ((wb >> (ce * LPE)) & 1ull) {
          const uint32_t* s = (const uint32_t*)(sgrid + ce * GS + part * 16);
          uint4 v; v.x = s[0]; v.y = s[1]; v.z = s[2]; v.w = s[3];
          live[c] = v;
        }
      }
    }
  }
  if (P.seg_count) {
    // one refill request per env that took a spare (however many): the generator draws head - tail episodes for it.
    // live_gen (DynamicObstacles): the request is "this env's episode ended", served in place before the next step.
    const bool want = active && lead && (P.live_gen ? ((a.flags & FLAG_RESET_PENDING) != 0u && P.phase == PHASE_STEP) : (h != h_in));
    const unsigned long long m = __ballot(want);
    if (m) {
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (want && qn + rank < (uint32_t)P.seg_cap) P.seg[(size_t)wg * P.seg_cap + qn + rank] = (uint32_t)e;
      qn = min(qn + (uint32_t)__popcll(m), (uint32_t)P.seg_cap);
      if (lane == 0) P.seg_count[wg] = qn;
    }
  }
  if (fin_total && lane == 0) atomicAdd(&P.counters[STAT_EPISODES + wg], (unsigned long long)fin_total);
}


}  // namespace mg
