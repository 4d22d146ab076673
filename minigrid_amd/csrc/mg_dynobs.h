// mg_dynobs.h — DynamicObstaclesEnv inside the fused step loop (round 4): the part of its step() and reset() that draws on the env's stream.
//
// The level's step() moves every obstacle BEFORE MiniGridEnv.step runs (dynamicobstacles.py:141-157: place_obj(obstacle, top = old - (1, 1),
// size = (3, 3), max_tries = 100) on the env's np_random, a failure swallowed), and its reset() places the agent and the obstacles on the same
// stream (:110-134).  A spare episode can therefore not be drawn ahead (the stream position of a reset depends on every step before it), which
// kept the level out of the fused kernel: rounds 1-3 ran a step as three launches (live redraw of the finished envs, k_move_obstacles, the
// step kernel) and could not fuse steps at all (0.91 G env-steps/s at 65 536 x 16x16).  Here ONE lane owns an env's stream (Pcg64Stream /
// PhiloxStream, mg_rng.h, in registers for the whole launch) and runs both -- the moves of a step, or the whole reset of an env whose
// episode ended -- as ONE loop of placement tries: a lane whose try is accepted moves on to its next obstacle by itself, so a wavefront runs
// as long as its unluckiest lane needs in TOTAL, not per obstacle (an acceptance loop per obstacle would make all 64 lanes wait for the
// unluckiest lane eight times a step), and the few lanes that redraw an episode do it under the same loop as the moves of the others.
//
// Host-callable (MG_HD): mg_selftest_dynobs runs it on the CPU against the oracle (tests/test_abi_cpu.py), so the draw order -- the one thing
// a GPU run cannot localise -- is pinned without a device.
#pragma once
#include "mg_device.h"
#include "mg_rng.h"
#include <type_traits>

namespace mg {

// g      the env's row-major byte grid (index y * W + x).  regen: already holds the level's constant part -- the outer wall and the goal
//        (dynamicobstacles.py:112-119; the caller copies it from a template) -- and no obstacle.
// regen  MiniGridEnv.reset's _gen_grid from the agent on (:121-132): the agent at its fixed start (sx >= 0) or place_agent() (top = (0, 0),
//        size = the grid, no try limit, then _rand_int(0, 4) for the direction; minigrid_env.py:383-395), then n times
//        place_obj(Ball(), max_tries = 100) over the whole grid.  A failure there is the reference's RecursionError out of reset(): `failed`.
// move   the obstacle moves of one step (:146-157), in list order; an obstacle that finds no place in 101 tries stays.
// obst   byte i = cell index of obstacle i, in list order (GenResult.aux, mg_gen.h).  w_magic = ceil(2^16 / W) (cell index -> row).
// place_obj's try (minigrid_env.py:339-363): `if num_tries > max_tries: raise`, then the try is counted; x then y are drawn
// (_rand_int = integers(low, high): one bounded 32-bit Lemire draw each, none when the range is a single value -- rand_int, mg_rng.h); a cell
// that holds anything, or the agent's cell ((-1, -1) while the agent itself is placed), is rejected.
// One placement try's position: x = _rand_int(topx, hx), then y = _rand_int(topy, hy) (minigrid_env.py:347-350).
// numpy's PCG64 hands out the two halves of a 64-bit output as two 32-bit draws (has32 / cache32, mg_rng.h), so a try costs exactly ONE state
// step whichever half the stream stands at: stepped once here, the two words picked by has32 -- no `if (has32)` around each draw (64 lanes at
// mixed phases would walk both sides of both).  Lemire's bounded draw accepts a word unless the low half of word * range is below the range
// (probability range / 2^32): such a try -- or a range of one value, which draws nothing -- is redone from the saved state by the general code.
// (REDO: every try takes the redo path -- the host self-test's way of reaching it.)
template <class R, bool REDO = false>
MG_HD void dynobs_draw_xy(R& rng, int topx, int hx, int topy, int hy, int& x, int& y) {
  if constexpr (std::is_same<R, Pcg64Stream>::value) {
    const uint32_t rx = (uint32_t)(hx - topx), ry = (uint32_t)(hy - topy);
    const Pcg64Stream saved = rng;
    const uint64_t n64 = rng.next64();
    const uint32_t lo = (uint32_t)n64, hi = (uint32_t)(n64 >> 32);
    const uint32_t w0 = rng.has32 ? rng.cache32 : lo, w1 = rng.has32 ? lo : hi;
    const uint64_t m0 = (uint64_t)w0 * rx, m1 = (uint64_t)w1 * ry;
    rng.cache32 = hi;                    // (has32 as it was: two draws later the stream stands at the same half)
    x = topx + (int)(m0 >> 32); y = topy + (int)(m1 >> 32);
    if (__builtin_expect(REDO || rx < 2u || ry < 2u || (uint32_t)m0 < rx || (uint32_t)m1 < ry, 0)) {
      rng = saved;
      x = rand_int(rng, topx, hx);
      y = rand_int(rng, topy, hy);
    }
  } else {
    x = rand_int(rng, topx, hx);
    y = rand_int(rng, topy, hy);
  }
}

// The loop body is straight-line but for its two stores and the agent's direction draw: a lane's try is accepted or not, its job advances or
// not, by selects -- an if / continue chain cost ~350 instructions per try, half of them exec-mask bookkeeping (profiles/r4/dynobs_attr_first.txt).
template <class R, bool REDO = false>
MG_HD void dynobs_place(R& rng, uint8_t* g, int W, int H, uint32_t w_magic, int n, bool regen, bool move, int sx, int sy, int sdir,
                        uint32_t& ax, uint32_t& ay, uint32_t& adir, uint64_t& obst, bool& failed, bool& changed) {
  int i = n;                               // the lane's current job: -1 = the agent (random start), 0 .. n-1 = obstacle i, n = nothing left
  int px = (int)ax, py = (int)ay;          // the agent position place_obj must avoid
  uint32_t pd = adir;
  uint64_t o = obst;
  if (regen) {
    o = 0;
    if (sx >= 0) { px = sx; py = sy; pd = (uint32_t)sdir; i = 0; }
    else { px = -1; py = -1; i = -1; }
  } else if (move) i = 0;
  int tries = 0;
  while (i < n) {
    tries++;                               // (`if num_tries > max_tries: raise` sits before the count: see `giveup` below)
    const bool agent_job = i < 0;
    const int sh = agent_job ? 0 : 8 * i;
    const int old = (int)((o >> sh) & 0xFFull);
    const int oy = (int)(((uint32_t)old * w_magic) >> 16), ox = old - oy * W;
    // a move: top = (max(x - 1, 0), max(y - 1, 0)), size (3, 3) clipped to the grid; a reset: the whole grid
    const int mtx = ox > 0 ? ox - 1 : 0, mty = oy > 0 ? oy - 1 : 0;
    const int topx = regen ? 0 : mtx, topy = regen ? 0 : mty;
    const int hx = regen ? W : (mtx + 3 < W ? mtx + 3 : W), hy = regen ? H : (mty + 3 < H ? mty + 3 : H);
    int x, y;
    dynobs_draw_xy<R, REDO>(rng, topx, hx, topy, hy, x, y);
    const int k = y * W + x;
    const bool ok = (uint32_t)g[k] == (uint32_t)CELL_EMPTY && !(x == px && y == py);
    const bool place = ok && !agent_job, agent_ok = ok && agent_job;
    if (place) {
      g[k] = (uint8_t)CELL_BALL_BLUE;
      if (!regen) g[old] = (uint8_t)CELL_EMPTY;
    }
    o = place ? ((o & ~(0xFFull << sh)) | ((uint64_t)(uint32_t)k << sh)) : o;
    changed |= place;
    if (agent_ok) {                        // place_agent: the position, then the direction
      px = x; py = y;
      pd = (uint32_t)rand_int(rng, 0, 4);
    }
    // the 101st try of an obstacle failed: the next loop head raises RecursionError("rejection sampling failed in place_obj") -- a move swallows
    // it (`except Exception: pass`: the obstacle stays), reset() does not.  (place_agent has no try limit: the reference would spin for ever on a
    // grid without a free cell -- there is always one here; the bound only keeps a kernel that was handed a corrupted grid from hanging the
    // device, and is reported like a failed reset.)
    const bool giveup = !ok && (agent_job ? tries > (1 << 16) : tries > 100);
    failed |= giveup && (regen || agent_job);
    if (giveup && agent_job) { px = 1; py = 1; }
    i = (agent_ok || (giveup && agent_job)) ? 0 : (place || giveup) ? i + 1 : i;
    tries = (ok || giveup) ? 0 : tries;
  }
  if (regen) { ax = (uint32_t)px; ay = (uint32_t)py; adir = pd; }
  obst = o;
}

// the constant part of a DynamicObstacles grid: Grid(W, H) + wall_rect(0, 0, W, H) + Goal at (W - 2, H - 2) (dynamicobstacles.py:112-119); 0 past W * H
MG_HD uint32_t dynobs_template_cell(int k, int W, int H, uint32_t w_magic) {
  if (k >= W * H) return 0u;
  const int y = (int)(((uint32_t)k * w_magic) >> 16), x = k - y * W;
  if (x == 0 || y == 0 || x == W - 1 || y == H - 1) return (uint32_t)CELL_WALL_GREY;
  return (x == W - 2 && y == H - 2) ? (uint32_t)CELL_GOAL : (uint32_t)CELL_EMPTY;
}

}  // namespace mg
