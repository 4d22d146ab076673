// mg_gen.h — device-side map generators (the reference's `_gen_grid` implementations on the hot path).
// One WAVEFRONT generates one episode into a byte grid in LDS, drawing from the env's stream in exactly the order
// the reference draws, so that with the PCG64 stream the layout for a given seed is the reference's layout.
// All control flow below is wave-uniform (draws come out of v_readlane, cells out of v_readfirstlane); the 64 lanes
// are used for the bulk parts: the jumped-ahead draws (mg_rng.h), clearing the grid, the reachability bitboards.
#pragma once
#include "mg_device.h"
#include "mg_rng.h"
#include "mg_dynobs.h"      // dynobs_draw_xy: one PCG64 step per pair of bounded draws (gen_goto_lane)

namespace mg {

// -DMG_GEN_ATTR builds only (profiles/variant_build.py genattr --units=<generator units>,mg_api.hip -DMG_GEN_ATTR; never the product library):
// per-phase cycle attribution of the wave-cooperative generators.  MG_GA(out, k) charges the s_memtime cycles since the previous mark to phase k
// (wave-uniform: SGPRs); generate_one adds the sums of every generated episode to counters[4 + k] (mg_debug_stamps).  Phases:
//   0 prologue (stream load, draw-buffer refills)   1 room lattice + door offsets   2 agent placement   3 connect_all
//   4 object placement (distractors, locked room)   5 reachability flood   6 mission / instruction drawing + validation
//   7 epilogue (stream position, stores)   8 MultiRoom: room-chain search   9 MultiRoom: walls + doors
//   [10] attempts (whole-level tries), [11] episodes -- counts, not cycles
#if defined(MG_GEN_ATTR) && defined(__HIP_DEVICE_COMPILE__)
constexpr int MG_GA_N = 10;
struct GenAttr { uint64_t t; uint64_t ph[MG_GA_N]; uint32_t attempts; };
#define MG_GA(out, k) do { const uint64_t now_ = __builtin_readcyclecounter(); (out).ga.ph[k] += now_ - (out).ga.t; (out).ga.t = now_; } while (0)
#define MG_GA_ATTEMPT(out) do { (out).ga.attempts++; } while (0)
#else
#define MG_GA(out, k) do { } while (0)
#define MG_GA_ATTEMPT(out) do { } while (0)
#endif

struct GenParams {
  int kind, W, H;
  int start_x, start_y, start_dir;   // Empty
  int num_crossings, obstacle_cell;  // Crossing (obstacle_cell = cell code of Lava()/Wall())
  int num_dists;                     // GoToRedBall
  int strip2_row;                    // DistShift
  int room_size;                     // RoomGrid levels
  int random_length;                 // Memory
  int max_steps;                     // sentence levels with a fixed step limit
  int instr_off;                     // sentence levels: byte offset from the wave's grid to INSTR_WORDS u64 of LDS for the instruction record
  int scratch_off;                   // byte offset from the wave's grid to GEN_SCRATCH_BYTES of LDS that outlive a checkpoint restart
};

struct GenResult {
  uint32_t ax, ay, dir, mission;
  uint64_t aux;       // per-env auxiliary word: DynamicObstacles: byte i = cell index (y*W+x) of obstacle i, in list
                      // order; BabyAI GoTo levels: bitboard (bit y*W+x) of GoToInstr's tracked object positions
  uint32_t gstate;    // LevelGen: locked_room as this episode leaves it (generator state carried from episode to episode)
  uint32_t carry;     // cell code the agent starts with in its hands (PutNext start_carrying), 0 = nothing
  uint32_t retries;   // whole-map regenerations (RejectSampling / RecursionError in the reference)
  bool failed;        // retry bound exhausted
  uint32_t stuck;     // RoomGrid.place_agent's endless loop was met (room_stuck); set by generate_one to 0 before the first pass
  uint32_t resume;    // set by generate_one: this pass restarts from the generator's last checkpoint (state in the wave's scratch words)
#if defined(MG_GEN_ATTR) && defined(__HIP_DEVICE_COMPILE__)
  GenAttr ga;
#endif
};

// Byte grid owned by one wave, row-major index y*W+x (core/grid.py:28-35,65-78).  get/set take wave-uniform
// coordinates: every lane reads (broadcast) or writes (same value) the same LDS byte.
struct GridRef {
  static constexpr bool kWave = true;
  uint8_t* p; int W, H; int lane;
  MG_D uint32_t get(int x, int y) const { return uni32((uint32_t)p[y * W + x]); }
  MG_D void set(int x, int y, uint32_t c) { p[y * W + x] = (uint8_t)c; }
  // Grid(width,height) + wall_rect(0,0,W,H) (core/grid.py:28-35,104-108), 64 cells per instruction
  MG_D void clear_with_walls() {
    MG_WAVE_LDS_SYNC();
    const bool edge_x = lane == 0 || lane == W - 1;
    for (int y = 0; y < H; y++)                      // one row per instruction (W <= 25 lanes busy), no division
      if (lane < W) p[y * W + lane] = (uint8_t)((edge_x || y == 0 || y == H - 1) ? CELL_WALL_GREY : CELL_EMPTY);
    MG_WAVE_LDS_SYNC();
  }
  // Grid(width, height) alone: every cell None (core/grid.py:28-35)
  MG_D void clear_empty() {
    MG_WAVE_LDS_SYNC();
    for (int y = 0; y < H; y++)
      if (lane < W) p[y * W + lane] = (uint8_t)CELL_EMPTY;
    MG_WAVE_LDS_SYNC();
  }
  // RoomGrid._gen_grid's wall lattice (roomgrid.py:123-150): a wall on every st-th row and column, None elsewhere.  One row per instruction;
  // the row phase is a counter and the column test is taken once (a `y % st` in the loop is a scalar division per row: this loop was 6-7 % of a
  // maze episode, profiles/r6/refill_attribution_goto_before.txt)
  MG_D void lattice(int st) {
    MG_WAVE_LDS_SYNC();
    const bool colwall = (lane % st) == 0;
    uint8_t* q = p + lane;
    int ph = 0;
    for (int y = 0; y < H; y++) {
      if (lane < W) *q = (uint8_t)((colwall || ph == 0) ? CELL_WALL_GREY : CELL_EMPTY);
      q += W;
      if (++ph == st) ph = 0;
    }
    MG_WAVE_LDS_SYNC();
  }
  // 8x8 grids only (bit index = y*8+x = lane): cells the reachability flood may pass (None or any door), cells
  // holding a non-wall object, and the number of red balls
  // grids of at most 64 cells (bit index = y*W+x = lane): cells the reachability flood may pass (None or any door),
  // cells holding a non-wall object, and the cells holding exactly `desc`
  MG_D void reach_masks(uint64_t& passable, uint64_t& objects, uint32_t desc, uint64_t& matches) const {
    MG_WAVE_LDS_SYNC();
    const bool in = lane < W * H;
    const uint32_t c = in ? (uint32_t)p[lane] : (uint32_t)CELL_WALL_GREY, t = cell_type(c);
    const bool pass = c == CELL_EMPTY || t == T_DOOR || t == T_DOOR_CLOSED || t == T_DOOR_LOCKED;
    passable = __ballot(in && pass);
    objects = __ballot(in && !pass && t != T_WALL);
    matches = __ballot(in && c == desc);
  }
};

// (Host-callable -- MG_HD -- like the generators templated on the grid type below: mg_selftest_generate runs them on the CPU against the oracle.)
// The same grid for ONE LANE (k_refill_lane, mg_genlane.h: a lane draws a whole episode by itself, 64 episodes per wavefront): the lane's
// private byte grid in LDS, coordinates per lane.  For grids of at most 64 cells it also keeps the board of non-empty cells up to date,
// so that reach_masks (two calls per GoToRedBall attempt) visits the handful of objects instead of scanning every cell.
struct LaneGrid {
  static constexpr bool kWave = false;
  uint8_t* p; int W, H; int lane;
  uint64_t nonempty;                 // bit y*W+x: the cell is not None (valid while W*H <= 64)
  uint64_t walls;                    // ... the outer wall ring clear_with_walls drew
  MG_HD uint32_t get(int x, int y) const { return (uint32_t)p[y * W + x]; }
  MG_HD void set(int x, int y, uint32_t c) {
    const int k = y * W + x;
    p[k] = (uint8_t)c;
    if (W * H <= 64) nonempty = c == CELL_EMPTY ? nonempty & ~(1ull << k) : nonempty | (1ull << k);
  }
  MG_HD void clear_with_walls() {
    if (W == 8 && H == 8 && ((uintptr_t)p & 3u) == 0u) {
      // the 8 x 8 room of the single-room levels as sixteen dword stores (round 6; the byte loop below is ~600 instructions per lane and episode)
      uint32_t* q = (uint32_t*)p;
      const uint32_t w4 = CELL_WALL_GREY * 0x01010101u, e4 = CELL_EMPTY * 0x01010101u;
      const uint32_t left = (e4 & 0xFFFFFF00u) | (uint32_t)CELL_WALL_GREY, right = (e4 & 0x00FFFFFFu) | ((uint32_t)CELL_WALL_GREY << 24);
      q[0] = w4; q[1] = w4; q[14] = w4; q[15] = w4;
#pragma unroll
      for (int y = 1; y < 7; y++) { q[2 * y] = left; q[2 * y + 1] = right; }
      nonempty = walls = 0xFF818181818181FFull;
      return;
    }
    nonempty = 0;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const bool wall = x == 0 || x == W - 1 || y == 0 || y == H - 1;
        p[y * W + x] = (uint8_t)(wall ? CELL_WALL_GREY : CELL_EMPTY);
        if (wall && W * H <= 64) nonempty |= 1ull << (y * W + x);
      }
    walls = nonempty;
  }
  MG_HD void clear_empty() {
    nonempty = 0; walls = 0;
    for (int k = 0; k < W * H; k++) p[k] = (uint8_t)CELL_EMPTY;
  }
  MG_HD void reach_masks(uint64_t& passable, uint64_t& objects, uint32_t desc, uint64_t& matches) const {
    const uint64_t all = (W * H >= 64) ? ~0ull : ((1ull << (W * H)) - 1ull);
    uint64_t pa = ~nonempty & all, ob = 0, ma = 0;
    uint64_t m = nonempty & ~walls;                 // the outer walls are neither passable nor objects and match no description
    while (m) {
      const int k = __builtin_ffsll((long long)m) - 1;
      const uint64_t bit = 1ull << k;
      m &= m - 1ull;
      const uint32_t c = (uint32_t)p[k], t = cell_type(c);
      const bool pass = t == T_DOOR || t == T_DOOR_CLOSED || t == T_DOOR_LOCKED;
      pa |= pass ? bit : 0ull;
      ob |= (!pass && t != T_WALL) ? bit : 0ull;
      ma |= c == desc ? bit : 0ull;
    }
    passable = pa; objects = ob; matches = ma;
  }
};

// ---- lane-parallel draws of the wave form (round 6) ----
// Lane-parallel Lemire draw out of the draw buffer: this lane's value for logical draw p and a range of r >= 2 values, and whether the draw is
// SAFE -- numpy's rejection step cannot apply to it (buffered_bounded_lemire_uint32 looks at its threshold only when leftover < r, and the
// threshold of a power-of-two range is 0).  An unsafe draw is never used speculatively: the caller falls back to the scalar rand_int there.
template <class R>
MG_D uint32_t peek_bounded(const R& rng, uint32_t p, uint32_t r, bool& safe) {
  const uint64_t m = (uint64_t)rng.peek_lane(p) * r;
  safe = (uint32_t)m >= r || (r & (r - 1u)) == 0u;
  return (uint32_t)(m >> 32);
}
// MiniGridEnv.place_obj (minigrid_env.py:313-372).  (ax, ay) is the agent position to avoid ((-1,-1) while the
// agent itself is being placed).  near_reject = core/roomgrid.py:11-20 reject_next_to.  max_tries < 0 = math.inf.
// Returns false on the reference's RecursionError.
template <class R, class G>
MG_HD bool place_obj(R& rng, G& g, uint32_t cell, int topx, int topy, int sx, int sy, int ax, int ay,
                    bool near_reject, int max_tries, int& px, int& py) {
  topx = topx < 0 ? 0 : topx; topy = topy < 0 ? 0 : topy;
  const int hx = min(topx + sx, g.W), hy = min(topy + sy, g.H);
  int tries = 0;
  // Speculative rejection sampling: a try consumes exactly two draws as long as numpy's Lemire step does not reject one of them (never for a
  // power-of-two range: threshold (2^32 - r) % r == 0; otherwise only when the draw's low product word is below the range -- peek_bounded's
  // `safe`), so lane t can evaluate try t from draws wpos+2t, wpos+2t+1 and the first acceptable lane is the reference's accepted try.  The
  // tries before the first unsafe one are evaluated here; that one is taken by the scalar loop below.  (Round 6: any range -- rooms whose size
  // is not a power of two, MultiRoom's, placed their objects one scalar try at a time.)
  const uint32_t rx = (uint32_t)(hx - topx), ry = (uint32_t)(hy - topy);
  if constexpr (G::kWave) if (rx >= 2u && ry >= 2u && (int)rx > 0 && (int)ry > 0 && !rng.dead()) {
    const uint32_t win = rng.window();
    const uint32_t p0 = rng.wpos + 2u * (uint32_t)g.lane;
    const bool valid = p0 + 1u < win;
    bool sa, sb;
    const uint32_t vx = peek_bounded(rng, p0, rx, sa), vy = peek_bounded(rng, p0 + 1u, ry, sb);
    const int x = topx + (int)vx, y = topy + (int)vy;
    const unsigned long long unsafe = __ballot(valid && !(sa && sb));
    const int nhave = win > rng.wpos ? (int)min((win - rng.wpos) >> 1, 64u) : 0;
    const int nvalid = unsafe ? min(nhave, __ffsll((long long)unsafe) - 1) : nhave;
    const uint32_t c = g.lane < nvalid ? (uint32_t)g.p[y * g.W + x] : 0u;
    const bool ok = g.lane < nvalid && c == CELL_EMPTY && !(x == ax && y == ay) && !(near_reject && (abs(ax - x) + abs(ay - y)) < 2);
    const unsigned long long m = __ballot(ok);
    if (m) {
      const int t = __ffsll((long long)m) - 1;
      if (max_tries >= 0 && t > max_tries) return false;
      rng.wpos += 2u * (uint32_t)(t + 1);
      px = (int)lane32((uint32_t)x, (uint32_t)t); py = (int)lane32((uint32_t)y, (uint32_t)t);
      if (cell != CELL_EMPTY) g.set(px, py, cell);
      return true;
    }
    tries = nvalid; rng.wpos += 2u * (uint32_t)nvalid;     // every evaluated try was rejected: carry on one by one
  }
  for (;;) {
    if (max_tries >= 0 && tries > max_tries) return false;
    if (rng.dead()) return false;                 // out of buffered draws: the caller replays with a larger budget
    tries++;
    int x = rand_int(rng, topx, hx);
    int y = rand_int(rng, topy, hy);
    if (g.get(x, y) != CELL_EMPTY) continue;
    if (x == ax && y == ay) continue;
    if (near_reject && (abs(ax - x) + abs(ay - y)) < 2) continue;
    if (cell != CELL_EMPTY) g.set(x, y, cell);
    px = x; py = y;
    return true;
  }
}
// MiniGridEnv.place_agent (minigrid_env.py:383-395)
template <class R, class G>
MG_HD bool place_agent(R& rng, G& g, int topx, int topy, int sx, int sy, int max_tries, GenResult& out) {
  int x, y;
  if (!place_obj(rng, g, CELL_EMPTY, topx, topy, sx, sy, -1, -1, false, max_tries, x, y)) return false;
  out.ax = (uint32_t)x; out.ay = (uint32_t)y;
  out.dir = (uint32_t)rand_int(rng, 0, 4);
  return true;
}

// envs/empty.py:97-114
template <class R, class G>
MG_HD void gen_empty(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  g.set(g.W - 2, g.H - 2, CELL_GOAL);
  if (P.start_x >= 0) { out.ax = P.start_x; out.ay = P.start_y; out.dir = P.start_dir; }
  else if (!place_agent(rng, g, 0, 0, g.W, g.H, -1, out)) out.failed = true;
  out.mission = 0;
}

// envs/doorkey.py:74-99
template <class R, class G>
MG_HD void gen_doorkey(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  g.set(g.W - 2, g.H - 2, CELL_GOAL);
  int split = rand_int(rng, 2, g.W - 2);
  for (int y = 0; y < g.H; y++) g.set(split, y, CELL_WALL_GREY);
  if (!place_agent(rng, g, 0, 0, split, g.H, -1, out)) out.failed = true;
  int door = rand_int(rng, 1, g.W - 2);
  g.set(split, door, make_cell(T_DOOR_LOCKED, C_YELLOW));
  int kx, ky;
  if (!place_obj(rng, g, make_cell(T_KEY, C_YELLOW), 0, 0, split, g.H, (int)out.ax, (int)out.ay, false, -1, kx, ky))
    out.failed = true;
  out.mission = 0;
}

// envs/crossing.py:131-188.  Rivers sit on even coordinates 2..size-3, at most 8 of them for the registered sizes
// (S9: 3+3, S11: 4+4), so the shuffled river list is 8 bytes packed in a u64 (byte = orientation << 7 | position)
// and the per-orientation sorted position lists are bitmasks.
template <class R, class G>
MG_HD void gen_crossing(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H;
  g.clear_with_walls();
  out.ax = 1; out.ay = 1; out.dir = 0;
  g.set(W - 2, H - 2, CELL_GOAL);
  uint64_t rivers = 0; int n = 0;
#pragma unroll 1
  for (int i = 2; i < H - 2; i += 2) { rivers |= (uint64_t)(i) << (8 * n); n++; }          // (v, i)
#pragma unroll 1
  for (int j = 2; j < W - 2; j += 2) { rivers |= (uint64_t)(0x80 | j) << (8 * n); n++; }   // (h, j)
  // np_random.shuffle(rivers): for i = n-1..1: j = random_interval(i); swap
#pragma unroll 1
  for (int i = n - 1; i >= 1; i--) {
    int j = (int)rand_interval(rng, (uint32_t)i);
    uint64_t bi = (rivers >> (8 * i)) & 0xFF, bj = (rivers >> (8 * j)) & 0xFF;
    rivers &= ~((0xFFull << (8 * i)) | (0xFFull << (8 * j)));
    rivers |= (bj << (8 * i)) | (bi << (8 * j));
  }
  n = min(n, P.num_crossings);
  uint32_t vmask = 0, hmask = 0; int nv = 0, nh = 0;    // sorted(rivers_v), sorted(rivers_h) as position bitmasks
#pragma unroll 1
  for (int k = 0; k < n; k++) {
    uint32_t b = (uint32_t)(rivers >> (8 * k)) & 0xFF;
    if (b & 0x80) { hmask |= 1u << (b & 0x7F); nh++; } else { vmask |= 1u << b; nv++; }
  }
#pragma unroll 1
  for (int p = 2; p < 32; p += 2) {
    if ((hmask >> p) & 1) {
#pragma unroll 1
      for (int i = 1; i < W - 1; i++) g.set(i, p, P.obstacle_cell); }
    if ((vmask >> p) & 1) {
#pragma unroll 1
      for (int j = 1; j < H - 1; j++) g.set(p, j, P.obstacle_cell); }
  }
  // path = [h]*len(rivers_v) + [v]*len(rivers_h); shuffle.  bit k of `path` = 1 for h.
  uint32_t path = (1u << nv) - 1u; const int np = nv + nh;
#pragma unroll 1
  for (int i = np - 1; i >= 1; i--) {
    int j = (int)rand_interval(rng, (uint32_t)i);
    uint32_t bi = (path >> i) & 1u, bj = (path >> j) & 1u;
    path = (path & ~((1u << i) | (1u << j))) | (bj << i) | (bi << j);
  }
  // limits_v = [0] + rivers_v + [H-1]; limits_h = [0] + rivers_h + [W-1]; walk the path opening one gap per river
  int room_i = 0, room_j = 0;
  int lv_lo = 0, lh_lo = 0;                 // limits_v[room_i], limits_h[room_j]
  uint32_t vrem = vmask, hrem = hmask;      // not-yet-crossed rivers, lowest bit = next limit
#pragma unroll 1
  for (int k = 0; k < np; k++) {
    int lv_hi = vrem ? __builtin_ctz(vrem) : H - 1;   // limits_v[room_i + 1]
    int lh_hi = hrem ? __builtin_ctz(hrem) : W - 1;   // limits_h[room_j + 1]
    int i, j;
    if ((path >> k) & 1u) {    // direction is h: cross the next vertical river
      i = lv_hi;
      j = rand_int(rng, lh_lo + 1, lh_hi);            // np_random.choice(range(a, b)) == a + integers(0, b-a)
      room_i++; lv_lo = lv_hi; vrem &= vrem - 1;
    } else {                   // direction is v: cross the next horizontal river
      i = rand_int(rng, lv_lo + 1, lv_hi);
      j = lh_hi;
      room_j++; lh_lo = lh_hi; hrem &= hrem - 1;
    }
    g.set(i, j, CELL_EMPTY);
  }
  (void)room_i; (void)room_j;
  out.mission = 0;
}

// BabyAI single-room GoTo levels on a 1x1 RoomGrid (core/roomgrid.py:123-179): envs/babyai/goto.py GoToRedBall
// 133-141 (+NoDists), GoToRedBallGrey 67-78, GoToRedBlueBall 661-677, GoToObj 256-260, GoToLocal 333-338;
// RoomGrid.place_agent (313-334), add_object/place_in_room (198-228,181-196: reject_next_to, max_tries=1000),
// add_distractors (396-438), check_objs_reachable (roomgrid_level.py:250-302) and the regenerate-on-reject loop
// (roomgrid_level.py:119-144).  Mission surface: verifier.py:73-103.  Grids of at most 64 cells: one 64-bit
// bitboard for the reachability flood and for GoToInstr's tracked positions (out.aux).
// Mission ids: red-ball levels 0 "go to the red ball" / 1 "go to a red ball"; GoToRedBlueBall 0 red / 1 blue;
// GoToObj / GoToLocal (article "a" ? 18 : 0) + COLOR_NAMES index * 3 + (key 0, ball 1, box 2).
enum : int { GOTO_REDBALL = 3, GOTO_REDBALLGREY = 16, GOTO_REDBLUEBALL = 17, GOTO_OBJ = 18, GOTO_LOCAL = 19 };
template <class R, class G>
MG_HD void gen_goto(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H, kind = P.kind;
  // column masks of the W x H bitboard (bit y*W+x)
  uint64_t col0 = 0, colL = 0, all = 0;
  for (int y = 0; y < H; y++) { col0 |= 1ull << (y * W); colL |= 1ull << (y * W + W - 1); }
  all = (W * H >= 64) ? ~0ull : ((1ull << (W * H)) - 1ull);
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();           // a rejected map needs nothing but the stream position: restart point
    g.clear_with_walls();
    // RoomGrid.place_agent: integers(0,1) for the room draws nothing; loop until the front cell is None or a wall
    bool ok = true;
    for (;;) {
      if (!place_agent(rng, g, 0, 0, W, H, 1000, out)) { ok = false; break; }
      uint32_t f = g.get((int)out.ax + dir_dx(out.dir), (int)out.ay + dir_dy(out.dir));
      if (f == CELL_EMPTY || cell_type(f) == T_WALL) break;
    }
    if (!ok) continue;
    int x, y;
    if (kind == GOTO_REDBALL || kind == GOTO_REDBALLGREY)
      if (!place_obj(rng, g, CELL_BALL_RED, 0, 0, W, H, (int)out.ax, (int)out.ay, true, 1000, x, y)) continue;
    // add_distractors: colour, type, place; remembered in list order (cell index / colour index / type index)
    uint64_t dpos = 0; uint32_t dcol = 0, dtyp = 0;
    bool red_or_blue_ball = false;
    const int ndist = kind == GOTO_OBJ ? 1 : min(P.num_dists, 8);
    for (int d = 0; d < ndist && ok; d++) {
      const uint32_t ci = (uint32_t)rand_int(rng, 0, 6);
      const uint32_t ti = (uint32_t)rand_int(rng, 0, 3);         // ["key", "ball", "box"] = 5, 6, 7
      const uint32_t color = color_from_sorted(ci);
      ok = place_obj(rng, g, make_cell(T_KEY + ti, color), 0, 0, W, H, (int)out.ax, (int)out.ay, true, 1000, x, y);
      dpos |= (uint64_t)(y * W + x) << (8 * d); dcol |= ci << (4 * d); dtyp |= ti << (4 * d);
      red_or_blue_ball |= ti == 1u && (color == C_RED || color == C_BLUE);
    }
    if (!ok) continue;
    uint32_t desc = CELL_BALL_RED;
    if (kind == GOTO_REDBALLGREY)                                // dist.color = "grey"
      for (int d = 0; d < ndist; d++) {
        const int idx = (int)((dpos >> (8 * d)) & 0xFF);
        g.set(idx % W, idx / W, make_cell(T_KEY + ((dtyp >> (4 * d)) & 15u), C_GREY));
      }
    if (kind == GOTO_REDBLUEBALL) {
      if (red_or_blue_ball) continue;                            // RejectSampling("can only have one blue or red ball")
      desc = make_cell(T_BALL, rand_int(rng, 0, 2) == 0 ? (uint32_t)C_RED : (uint32_t)C_BLUE);
      if (!place_obj(rng, g, desc, 0, 0, W, H, (int)out.ax, (int)out.ay, true, 1000, x, y)) continue;
    }
    uint64_t passable, objects, matches;
    if (kind != GOTO_OBJ) {
      // check_objs_reachable: flood from the agent through None/door cells; every non-wall object must be in the
      // visited set (= passable flood plus its 4-neighbourhood)
      g.reach_masks(passable, objects, desc, matches);
      uint64_t reach = 1ull << (out.ay * W + out.ax);
      for (;;) {
        const uint64_t grow = (((reach & ~colL) << 1) | ((reach & ~col0) >> 1) | (reach << W) | (reach >> W)) & all;
        const uint64_t next = reach | (grow & passable);
        if (next == reach) break;
        reach = next;
      }
      const uint64_t visited = (reach | ((reach & ~colL) << 1) | ((reach & ~col0) >> 1) | (reach << W) | (reach >> W)) & all;
      if (objects & ~visited) continue;           // RejectSampling("unreachable object")
    }
    uint32_t k = 0;
    if (kind == GOTO_LOCAL) k = (uint32_t)rand_int(rng, 0, ndist);                // _rand_elem(objs)
    if (kind == GOTO_OBJ || kind == GOTO_LOCAL) desc = make_cell(T_KEY + ((dtyp >> (4 * k)) & 15u), color_from_sorted((dcol >> (4 * k)) & 15u));
    g.reach_masks(passable, objects, desc, matches);
    out.aux = matches;                            // GoToInstr.reset_verifier -> desc.find_matching_objs: tracked positions
    const uint32_t many = __builtin_popcountll(matches) > 1 ? 1u : 0u;
    if (kind == GOTO_OBJ || kind == GOTO_LOCAL) out.mission = many * 18u + ((dcol >> (4 * k)) & 15u) * 3u + ((dtyp >> (4 * k)) & 15u);
    else if (kind == GOTO_REDBLUEBALL) out.mission = cell_color(desc) == C_BLUE ? 1u : 0u;
    else out.mission = many;                      // "go to the red ball" / "go to a red ball"
    return;
  }
  out.failed = true;
}

// gen_goto for ONE LANE (k_refill_lane / k_generate_lane: 64 episodes per wavefront) as ONE loop of draws -- round 6.  The literal form above
// runs a rejection loop per object: under SIMT every one of them lasts as long as its unluckiest lane (a try lands in the 8 x 8 room's free interior
// about every other time: the mean is two tries per object, the maximum over 64 lanes seven), nine objects one after the other, which is why the
// BabyAI-GoToRedBall refill ran best with a third of the lanes busy and still kept the chip busier than the step kernel it feeds (profiles/r6/
// kernel_stats_gotoredball_who_is_busy.txt).  Here a lane is a state machine -- job 0 the agent, then the red ball (GoToRedBall / -Grey), then the
// distractors, each a header (colour, type) and tries -- and every loop iteration takes exactly one PAIR of bounded draws (dynobs_draw_xy: one PCG64
// step whichever half of a 64-bit output the stream stands at): a lane whose try is accepted moves on by itself, so the wave runs as long as the lane
// with the largest TOTAL, not the sum of the per-object maxima.  Same draws in the same order as gen_goto (mg_selftest_generate against the oracle;
// the GPU goldens): header = rand_int(0, 6), rand_int(0, 3) (add_distractors, roomgrid.py:419-420), a try = x then y (place_obj,
// minigrid_env.py:347-350), the agent's direction after its accepted try (place_agent, :391-393).
template <class R>
MG_HD void gen_goto_lane(R& rng, LaneGrid& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H, kind = P.kind;
  uint64_t col0 = 0, colL = 0, all = 0;
  for (int y = 0; y < H; y++) { col0 |= 1ull << (y * W); colL |= 1ull << (y * W + W - 1); }
  all = (W * H >= 64) ? ~0ull : ((1ull << (W * H)) - 1ull);
  const int ndist = kind == GOTO_OBJ ? 1 : min(P.num_dists, 8);
  const int first_dist = (kind == GOTO_REDBALL || kind == GOTO_REDBALLGREY) ? 2 : 1;        // job 0 the agent, job 1 the red ball (those two levels)
  const int njobs = first_dist + ndist;
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    g.clear_with_walls();
    int j = 0, tries = 0;
    bool hdr = false, ok = true;
    int ax = -1, ay = -1;
    uint32_t adir = 0, ci = 0, ti = 0, dcol = 0, dtyp = 0;
    uint64_t dpos = 0;
    bool red_or_blue_ball = false;
    while (j < njobs) {
      const int d = j - first_dist;
      const bool need_hdr = d >= 0 && !hdr;
      int v0, v1;
      dynobs_draw_xy(rng, 0, need_hdr ? 6 : W, 0, need_hdr ? 3 : H, v0, v1);
      // a header: the distractor's colour and type; its tries follow
      ci = need_hdr ? (uint32_t)v0 : ci; ti = need_hdr ? (uint32_t)v1 : ti;
      // a try (place_obj: the count, the two draws, the cell, the agent's cell, reject_next_to)
      const bool try_ = !need_hdr;
      tries += try_ ? 1 : 0;
      const int k = try_ ? v1 * W + v0 : 0;
      const int dx = ax - v0, dy = ay - v1;
      const bool near = j > 0 && (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy) < 2;             // (the agent's own cell included; job 0 runs with the agent at (-1, -1) and no reject_fn)
      const bool acc = try_ && (uint32_t)g.p[k] == (uint32_t)CELL_EMPTY && !near;
      const bool giveup = try_ && !acc && tries > 1000;                                     // the 1001st rejected try: RecursionError -> the whole level again
      if (acc && j == 0) {
        // place_agent: the position, then the direction; RoomGrid.place_agent repeats it while the cell in front holds an object
        ax = v0; ay = v1;
        adir = (uint32_t)rand_int(rng, 0, 4);
        const uint32_t f = g.get(ax + dir_dx(adir), ay + dir_dy(adir));
        j = (f == CELL_EMPTY || cell_type(f) == T_WALL) ? 1 : 0;
      } else if (acc) {
        const uint32_t color = color_from_sorted(ci);
        const uint32_t cell = d < 0 ? (uint32_t)CELL_BALL_RED : make_cell(T_KEY + ti, color);
        g.set(v0, v1, cell);
        if (d >= 0) {
          dpos |= (uint64_t)(uint32_t)k << (8 * d); dcol |= ci << (4 * d); dtyp |= ti << (4 * d);
          red_or_blue_ball |= ti == 1u && (color == C_RED || color == C_BLUE);
        }
        j++;
      }
      hdr = acc ? false : (hdr || need_hdr);
      tries = acc ? 0 : tries;
      if (giveup) { ok = false; j = njobs; }
    }
    if (!ok) continue;
    out.ax = (uint32_t)ax; out.ay = (uint32_t)ay; out.dir = adir;
    int x, y;
    uint32_t desc = CELL_BALL_RED;
    if (kind == GOTO_REDBALLGREY)                                // dist.color = "grey"
      for (int d = 0; d < ndist; d++) {
        const int idx = (int)((dpos >> (8 * d)) & 0xFF);
        g.set(idx % W, idx / W, make_cell(T_KEY + ((dtyp >> (4 * d)) & 15u), C_GREY));
      }
    if (kind == GOTO_REDBLUEBALL) {
      if (red_or_blue_ball) continue;                            // RejectSampling("can only have one blue or red ball")
      desc = make_cell(T_BALL, rand_int(rng, 0, 2) == 0 ? (uint32_t)C_RED : (uint32_t)C_BLUE);
      if (!place_obj(rng, g, desc, 0, 0, W, H, ax, ay, true, 1000, x, y)) continue;
    }
    uint64_t passable, objects, matches;
    if (kind != GOTO_OBJ) {
      g.reach_masks(passable, objects, desc, matches);
      uint64_t reach = 1ull << (ay * W + ax);
      for (;;) {
        const uint64_t grow = (((reach & ~colL) << 1) | ((reach & ~col0) >> 1) | (reach << W) | (reach >> W)) & all;
        const uint64_t next = reach | (grow & passable);
        if (next == reach) break;
        reach = next;
      }
      const uint64_t visited = (reach | ((reach & ~colL) << 1) | ((reach & ~col0) >> 1) | (reach << W) | (reach >> W)) & all;
      if (objects & ~visited) continue;           // RejectSampling("unreachable object")
    }
    uint32_t kk = 0;
    if (kind == GOTO_LOCAL) kk = (uint32_t)rand_int(rng, 0, ndist);               // _rand_elem(objs)
    if (kind == GOTO_OBJ || kind == GOTO_LOCAL) desc = make_cell(T_KEY + ((dtyp >> (4 * kk)) & 15u), color_from_sorted((dcol >> (4 * kk)) & 15u));
    g.reach_masks(passable, objects, desc, matches);
    out.aux = matches;
    const uint32_t many = __builtin_popcountll(matches) > 1 ? 1u : 0u;
    if (kind == GOTO_OBJ || kind == GOTO_LOCAL) out.mission = many * 18u + ((dcol >> (4 * kk)) & 15u) * 3u + ((dtyp >> (4 * kk)) & 15u);
    else if (kind == GOTO_REDBLUEBALL) out.mission = cell_color(desc) == C_BLUE ? 1u : 0u;
    else out.mission = many;
    return;
  }
  out.failed = true;
}

// envs/lavagap.py:100-135
template <class R, class G>
MG_HD void gen_lavagap(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  out.ax = 1; out.ay = 1; out.dir = 0;
  g.set(g.W - 2, g.H - 2, CELL_GOAL);
  const int gx = rand_int(rng, 2, g.W - 2);
  const int gy = rand_int(rng, 1, g.H - 1);
  for (int j = 1; j < g.H - 1; j++) g.set(gx, j, (uint32_t)P.obstacle_cell);     // vert_wall(x, 1, height - 2, obstacle)
  g.set(gx, gy, CELL_EMPTY);
  out.mission = 0;
}

// envs/distshift.py:103-124
template <class R, class G>
MG_HD void gen_distshift(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  g.set(g.W - 2, 1, CELL_GOAL);
  for (int i = 0; i < g.W - 6; i++) { g.set(3 + i, 1, CELL_LAVA); g.set(3 + i, P.strip2_row, CELL_LAVA); }
  if (P.start_x >= 0) { out.ax = P.start_x; out.ay = P.start_y; out.dir = P.start_dir; }
  else if (!place_agent(rng, g, 0, 0, g.W, g.H, -1, out)) out.failed = true;
  out.mission = 0;
}

// envs/fourrooms.py:77-130 with agent_pos = goal_pos = None (the registered configuration)
template <class R, class G>
MG_HD void gen_fourrooms(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H;
  g.clear_with_walls();                                  // horz_wall(0,0), horz_wall(0,H-1), vert_wall(0,0), vert_wall(W-1,0)
  const int room_w = W / 2, room_h = H / 2;
  for (int j = 0; j < 2; j++) {
    for (int i = 0; i < 2; i++) {
      const int xL = i * room_w, yT = j * room_h, xR = xL + room_w, yB = yT + room_h;
      if (i + 1 < 2) {
        for (int k = 0; k < room_h; k++) g.set(xR, yT + k, CELL_WALL_GREY);
        g.set(xR, rand_int(rng, yT + 1, yB), CELL_EMPTY);
      }
      if (j + 1 < 2) {
        for (int k = 0; k < room_w; k++) g.set(xL + k, yB, CELL_WALL_GREY);
        g.set(rand_int(rng, xL + 1, xR), yB, CELL_EMPTY);
      }
    }
  }
  if (!place_agent(rng, g, 0, 0, W, H, -1, out)) out.failed = true;
  int x, y;
  if (!place_obj(rng, g, CELL_GOAL, 0, 0, W, H, (int)out.ax, (int)out.ay, false, -1, x, y)) out.failed = true;
  out.mission = 0;
}

// envs/fetch.py:107-160 (P.num_dists = numObjs <= 8).  Mission id = syntax*12 + COLOR_NAMES index*2 + (key 0 | ball 1):
// the step rule recovers the target (type, colour) from it, so no extra per-env state is needed.
template <class R, class G>
MG_HD void gen_fetch(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  uint64_t objs = 0;                           // byte k = colour index * 2 + type index of object k
  int x, y;
  const int n = min(P.num_dists, 8);
  for (int k = 0; k < n; k++) {
    const uint32_t ty = (uint32_t)rand_int(rng, 0, 2);          // _rand_elem(["key", "ball"])
    const uint32_t ci = (uint32_t)rand_int(rng, 0, 6);          // _rand_elem(COLOR_NAMES)
    if (!place_obj(rng, g, make_cell(ty == 0 ? (uint32_t)T_KEY : (uint32_t)T_BALL, color_from_sorted(ci)), 0, 0, g.W, g.H,
                   -1, -1, false, -1, x, y)) out.failed = true;
    objs |= (uint64_t)(ci * 2u + ty) << (8 * k);
  }
  if (!place_agent(rng, g, 0, 0, g.W, g.H, -1, out)) out.failed = true;
  const int t = rand_int(rng, 0, n);
  const uint32_t syntax = (uint32_t)rand_int(rng, 0, 5);
  out.mission = syntax * 12u + (uint32_t)((objs >> (8 * t)) & 0xFF);
}

// envs/gotoobject.py:93-135 (P.num_dists = numObjs <= 8).  Mission id = COLOR_NAMES index * 3 + (key 0 | ball 1 | box 2);
// out.aux = one-bit board of target_pos, which is a POSITION fixed at reset (the object may be carried away later).
template <class R, class G>
MG_HD void gen_gotoobject(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  uint64_t objs = 0, poss = 0;                 // byte k = colour index * 3 + type index / cell index of object k
  uint32_t used = 0;                           // bit (colour index * 3 + type index)
  int x, y;
  const int n = min(P.num_dists, 8);
#pragma unroll 1
  for (int k = 0; k < n && !rng.dead();) {
    const uint32_t ty = (uint32_t)rand_int(rng, 0, 3);          // _rand_elem(["key", "ball", "box"])
    const uint32_t ci = (uint32_t)rand_int(rng, 0, 6);          // _rand_elem(COLOR_NAMES)
    const uint32_t id = ci * 3u + ty;
    if ((used >> id) & 1u) continue;                            // gotoobject.py:112-113
    if (!place_obj(rng, g, make_cell((uint32_t)T_KEY + ty, color_from_sorted(ci)), 0, 0, g.W, g.H, -1, -1, false, -1, x, y)) out.failed = true;
    used |= 1u << id;
    objs |= (uint64_t)id << (8 * k);
    poss |= (uint64_t)(y * g.W + x) << (8 * k);
    k++;
  }
  if (!place_agent(rng, g, 0, 0, g.W, g.H, -1, out)) out.failed = true;
  const int t = rand_int(rng, 0, n);
  out.mission = (uint32_t)((objs >> (8 * t)) & 0xFF);
  out.aux = 1ull << ((poss >> (8 * t)) & 63u);
}

// envs/putnear.py:101-175 (P.num_dists = numObjs <= 8).  Objects with distinct (type, colour), none within Chebyshev
// distance 1 of an earlier one (place_obj's reject_fn near_obj; no max_tries); then the agent, the object to move and a
// different target object.  Mission id = ((move colour * 3 + move type) * 6 + target colour) * 3 + target type over
// COLOR_NAMES x [key, ball, box] (the order of putnear.py:72-80's placeholders); out.aux = one-bit board of target_pos.
template <class R, class G>
MG_HD void gen_putnear(R& rng, G& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  uint64_t objs = 0, poss = 0;                 // byte k = colour index * 3 + type index / cell index of object k
  uint32_t used = 0;
  const int n = min(P.num_dists, 8);
#pragma unroll 1
  for (int k = 0; k < n && !rng.dead();) {
    const uint32_t ty = (uint32_t)rand_int(rng, 0, 3);          // _rand_elem(types)
    const uint32_t ci = (uint32_t)rand_int(rng, 0, 6);          // _rand_elem(COLOR_NAMES)
    const uint32_t id = ci * 3u + ty;
    if ((used >> id) & 1u) continue;                            // putnear.py:130-131
    int x = 0, y = 0;
#pragma unroll 1
    for (;;) {                                                  // place_obj over the whole grid (agent_pos is (-1, -1) here)
      if (rng.dead()) break;
      x = rand_int(rng, 0, g.W); y = rand_int(rng, 0, g.H);
      if (g.get(x, y) != CELL_EMPTY) continue;
      bool near = false;
      for (int j = 0; j < k; j++) {
        const int q = (int)((poss >> (8 * j)) & 0xFF), px = q % g.W, py = q / g.W;
        near |= abs(x - px) <= 1 && abs(y - py) <= 1;
      }
      if (!near) break;
    }
    g.set(x, y, make_cell((uint32_t)T_KEY + ty, color_from_sorted(ci)));
    used |= 1u << id;
    objs |= (uint64_t)id << (8 * k);
    poss |= (uint64_t)(y * g.W + x) << (8 * k);
    k++;
  }
  if (!place_agent(rng, g, 0, 0, g.W, g.H, -1, out)) out.failed = true;
  const int mv = rand_int(rng, 0, n);
  int tg = mv;
  while (tg == mv && !rng.dead()) tg = rand_int(rng, 0, n);
  const uint32_t idm = (uint32_t)((objs >> (8 * mv)) & 0xFF), idt = (uint32_t)((objs >> (8 * tg)) & 0xFF);
  out.mission = idm * 18u + idt;               // ((cm * 3 + tm) * 6 + ct) * 3 + tt
  out.aux = 1ull << ((poss >> (8 * tg)) & 63u);
}

// envs/gotodoor.py:92-131.  The room is w x h <= W x H in the top-left corner; the rest of the grid stays None.
// Mission id = COLOR_NAMES index of the target door (door colours are distinct, so it identifies the door).
// (templated on the grid type like the generators above -- round 4: the per-lane form runs on the host against the oracle,
// tests/test_generators_cpu.py; the device's lane kernels do not serve this level yet)
template <class R, class G>
MG_HD void gen_gotodoor(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int w = rand_int(rng, 5, g.W + 1);
  const int h = rand_int(rng, 5, g.H + 1);
  if constexpr (G::kWave) {
  MG_WAVE_LDS_SYNC();
  for (int y = 0; y < g.H; y++)
    if (g.lane < g.W) {
      const bool in_room = g.lane < w && y < h;
      const bool edge = g.lane == 0 || g.lane == w - 1 || y == 0 || y == h - 1;
      g.p[y * g.W + g.lane] = (uint8_t)((in_room && edge) ? CELL_WALL_GREY : CELL_EMPTY);      // wall_rect(0, 0, w, h)
    }
  MG_WAVE_LDS_SYNC();
  } else {
    g.clear_empty();
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) if (x == 0 || x == w - 1 || y == 0 || y == h - 1) g.set(x, y, CELL_WALL_GREY);
  }
  int px[4], py[4];
  px[0] = rand_int(rng, 2, w - 2); py[0] = 0;
  px[1] = rand_int(rng, 2, w - 2); py[1] = h - 1;
  px[2] = 0; py[2] = rand_int(rng, 2, h - 2);
  px[3] = w - 1; py[3] = rand_int(rng, 2, h - 2);
  uint32_t colors = 0, used = 0;               // nibble k = COLOR_NAMES index of door k
  for (int n = 0; n < 4 && !rng.dead();) {
    const uint32_t ci = (uint32_t)rand_int(rng, 0, 6);
    if ((used >> ci) & 1u) continue;
    used |= 1u << ci; colors |= ci << (4 * n); n++;
  }
  for (int k = 0; k < 4; k++) g.set(px[k], py[k], make_cell(T_DOOR_CLOSED, color_from_sorted((colors >> (4 * k)) & 15u)));
  if (!place_agent(rng, g, 0, 0, w, h, -1, out)) out.failed = true;
  const int d = rand_int(rng, 0, 4);
  out.mission = (colors >> (4 * d)) & 15u;
}

// ---- core/roomgrid.py pieces for the 1 x 2 RoomGrid levels (Unlock, UnlockPickup, BlockedUnlockPickup) ----
// RoomGrid.place_agent's `while True` (roomgrid.py:327-332) has no bound: when the room holds a free cell but every free cell faces a
// non-wall object in all four directions, the reference never returns (BabyAI-SynthS5R2: 18 objects in six 3 x 3 rooms, about 0.4 % of
// the episodes).  Decided exactly, one lane per cell of the room's rs x rs rectangle, before anything is drawn.  (A room without any free
// cell is not this case: place_obj's 1000 tries run out -- RecursionError, which the callers' retry loops handle.)
template <class G>
MG_HD bool room_stuck(G& g, int topx, int topy, int rs) {
  if (rs > 8) return false;
  if constexpr (!G::kWave) {
    // (one lane: the same decision cell by cell)
    bool any_free = false, any_ok = false;
    for (int ly = 0; ly < rs; ly++) for (int lx = 0; lx < rs; lx++) {
      const int x = topx + lx, y = topy + ly;
      if (x >= g.W || y >= g.H || (uint32_t)g.p[y * g.W + x] != CELL_EMPTY) continue;
      any_free = true;
      for (int d = 0; d < 4; d++) {
        const int fx = x + dir_dx((uint32_t)d), fy = y + dir_dy((uint32_t)d);
        if (fx < 0 || fy < 0 || fx >= g.W || fy >= g.H) continue;
        const uint32_t f = (uint32_t)g.p[fy * g.W + fx];
        any_ok = any_ok || f == CELL_EMPTY || cell_type(f) == T_WALL;
      }
    }
    return any_free && !any_ok;
  } else {
  MG_WAVE_LDS_SYNC();
  const int ly = g.lane / rs, lx = g.lane - ly * rs, x = topx + lx, y = topy + ly;
  const bool in = ly < rs && x < g.W && y < g.H;
  const bool is_free = in && (uint32_t)g.p[y * g.W + x] == CELL_EMPTY;
  bool ok = false;
  if (is_free) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const int fx = x + dir_dx(d), fy = y + dir_dy(d);
      if (fx < 0 || fy < 0 || fx >= g.W || fy >= g.H) continue;
      const uint32_t f = (uint32_t)g.p[fy * g.W + fx];
      ok = ok || f == CELL_EMPTY || cell_type(f) == T_WALL;
    }
  }
  return __ballot(is_free) != 0ull && __ballot(ok) == 0ull;
  }
}
// RoomGrid.place_agent (roomgrid.py:313-334): place_agent in the room until the front cell is None or a wall
template <class R, class G>
MG_HD bool rg_place_agent(R& rng, G& g, int topx, int topy, int rs, GenResult& out) {
  if (room_stuck(g, topx, topy, rs)) { out.stuck = 1u; return false; }     // ends the attempt like a RecursionError; the episode is marked
  for (;;) {
    if (!place_agent(rng, g, topx, topy, rs, rs, 1000, out)) return false;
    const uint32_t f = g.get((int)out.ax + dir_dx(out.dir), (int)out.ay + dir_dy(out.dir));
    if (f == CELL_EMPTY || cell_type(f) == T_WALL) return true;
  }
}
// envs/unlock.py:75-88, envs/unlockpickup.py:82-97, envs/blockedunlockpickup.py:90-110 (variant 0 / 1 / 2)
template <class R, class G>
MG_HD void gen_unlock_family(R& rng, G& g, const GenParams& P, GenResult& out, int variant) {
  const int rs = P.room_size, W = g.W, H = g.H;
  // RoomGrid._gen_grid (roomgrid.py:123-179): wall_rect per room; one door position drawn per room pair; the agent
  // "starts in the middle": that provisional position is what reject_next_to / place_obj see until place_agent
  if constexpr (G::kWave) {
  MG_WAVE_LDS_SYNC();
  for (int y = 0; y < H; y++)
    if (g.lane < W) {
      const bool wall = y == 0 || y == H - 1 || (g.lane % (rs - 1)) == 0;
      g.p[y * W + g.lane] = (uint8_t)(wall ? CELL_WALL_GREY : CELL_EMPTY);
    }
  MG_WAVE_LDS_SYNC();
  } else {
    g.clear_empty();
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if (y == 0 || y == H - 1 || (x % (rs - 1)) == 0) g.set(x, y, CELL_WALL_GREY);
  }
  const int door_x = rs - 1, door_y = rand_int(rng, 1, rs - 1);          // room (0,0).door_pos[0] = (x_m, rand(y_l, y_m))
  const int mid_x = (2 / 2) * (rs - 1) + rs / 2, mid_y = rs / 2;         // provisional agent_pos (num_cols = 2, num_rows = 1)
  int x, y;
  uint32_t box_ci = 0;
  if (variant >= 1) {                                                    // add_object(1, 0, kind="box"): colour draw, then place
    box_ci = (uint32_t)rand_int(rng, 0, 6);
    if (!place_obj(rng, g, make_cell(T_BOX, color_from_sorted(box_ci)), rs - 1, 0, rs, rs, mid_x, mid_y, true, 1000, x, y)) out.failed = true;
  }
  const uint32_t door_ci = (uint32_t)rand_int(rng, 0, 6);                // add_door(0, 0, 0, locked=True): colour draw
  g.set(door_x, door_y, make_cell(T_DOOR_LOCKED, color_from_sorted(door_ci)));
  if (variant == 2) {                                                    // block the door with a ball of a random colour
    const uint32_t bc = (uint32_t)rand_int(rng, 0, 6);
    g.set(door_x - 1, door_y, make_cell(T_BALL, color_from_sorted(bc)));
  }
  // add_object(0, 0, "key", door.color)
  if (!place_obj(rng, g, make_cell(T_KEY, color_from_sorted(door_ci)), 0, 0, rs, rs, mid_x, mid_y, true, 1000, x, y)) out.failed = true;
  if (!rg_place_agent(rng, g, 0, 0, rs, out)) out.failed = true;
  out.mission = variant == 0 ? 0u : (variant == 1 ? box_ci : box_ci * 2u);
}

// (peek_bounded: above place_obj)
// Four bounded draws in a row (_rand_int over r0 .. r3 values, each >= 2): lanes 0..3 evaluate them in one pass.  false = not applicable here
// (too few buffered draws, or an unsafe draw): nothing was consumed, the caller draws them one by one.
template <class R, class G>
MG_D bool draw4_spec(R& rng, const G& g, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, int v[4]) {
  if (rng.dead() || rng.wpos + 4u > rng.window() || r0 < 2u || r1 < 2u || r2 < 2u || r3 < 2u) return false;
  const uint32_t l = (uint32_t)g.lane < 3u ? (uint32_t)g.lane : 3u;
  const uint32_t r = l == 0u ? r0 : l == 1u ? r1 : l == 2u ? r2 : r3;
  bool safe;
  const uint32_t val = peek_bounded(rng, rng.wpos + l, r, safe);
  if (__ballot(!safe)) return false;
  v[0] = (int)lane32(val, 0u); v[1] = (int)lane32(val, 1u); v[2] = (int)lane32(val, 2u); v[3] = (int)lane32(val, 3u);
  rng.wpos += 4u;
  return true;
}
// RoomGrid.connect_all's loop (roomgrid.py:362-394), speculatively: until a pick is accepted nothing the loop looks at changes, so lane t can
// evaluate iteration t -- the picks (i, j, k) from draws wpos + per * t .., per = the draws an iteration consumes -- against the CURRENT doors, and
// the first lane whose pick is a free wall between two unlocked rooms is the iteration the reference accepts; the lanes before it are its rejected
// iterations.  Returns the iterations consumed (their draws are consumed too); `acc`: the last of them was accepted, with its (i, j, k).  0 = nothing
// could be evaluated (no buffered draws left, or iteration 0 holds an unsafe draw): the caller runs one iteration the scalar way.
template <class R, class G>
MG_D int connect_spec(R& rng, const G& g, int nc, int nr, uint64_t doors, uint32_t locked, int max_iters, bool& acc, int& i, int& j, int& k) {
  acc = false;
  if (rng.dead()) return 0;
  const uint32_t di = nc > 1 ? 1u : 0u, dj = nr > 1 ? 1u : 0u, per = di + dj + 1u;      // (_rand_int over one value draws nothing)
  const uint32_t p0 = rng.wpos + per * (uint32_t)g.lane;
  const bool have = p0 + per <= rng.window() && g.lane < max_iters;
  bool s0 = true, s1 = true, s2 = true;
  const int li = di ? (int)peek_bounded(rng, p0, (uint32_t)nc, s0) : 0;
  const int lj = dj ? (int)peek_bounded(rng, p0 + di, (uint32_t)nr, s1) : 0;
  const int lk = (int)peek_bounded(rng, p0 + di + dj, 4u, s2);
  const unsigned long long unsafe = __ballot(have && !(s0 && s1 && s2));
  const int nhave = __popcll(__ballot(have));                                           // (the lanes that have their draws are a prefix)
  const int n = unsafe ? min(nhave, __ffsll((long long)unsafe) - 1) : nhave;
  if (n == 0) return 0;
  const bool nb = lk == 0 ? li < nc - 1 : lk == 1 ? lj < nr - 1 : lk == 2 ? li > 0 : lj > 0;
  const int r = lj * nc + li, r2 = r + (lk == 0 ? 1 : lk == 1 ? nc : lk == 2 ? -1 : -nc);
  const bool free_wall = nb && !((doors >> ((r * 4 + lk) & 63)) & 1ull);
  const bool unlocked = !(((locked >> (r & 31)) | (locked >> (r2 & 31))) & 1u);
  const unsigned long long okm = __ballot(g.lane < n && free_wall && unlocked);
  if (okm) {
    const int t = __ffsll((long long)okm) - 1;
    rng.wpos += per * (uint32_t)(t + 1);
    i = (int)lane32((uint32_t)li, (uint32_t)t); j = (int)lane32((uint32_t)lj, (uint32_t)t); k = (int)lane32((uint32_t)lk, (uint32_t)t);
    acc = true;
    return t + 1;
  }
  rng.wpos += per * (uint32_t)n;
  return n;
}

// RoomGrid.connect_all's find_reach (roomgrid.py:345-360) on the door nibbles (bit 4r+k: room r has a door on side k = right, down, left, up):
// the set of rooms reachable from `start` through doors, as bit 4r per room.  One pass = every reached room's four neighbours at once (a dozen
// 64-bit scalar operations); at most nrooms - 1 passes.  The reference recomputes the set in every iteration of connect_all's loop, rejected
// picks included; it is a function of the doors alone, so the callers recompute it only after a door was added (round 6: the per-room loops this
// replaces were 60-70 % of a maze episode's instructions, profiles/r6/refill_attribution_*.txt).
constexpr uint64_t ROOM_LSB = 0x1111111111111111ull;
MG_HD uint64_t rooms_reach(uint64_t doors, int start, int nc, int nrooms) {
  const uint64_t mR = doors & ROOM_LSB, mD = (doors >> 1) & ROOM_LSB, mL = (doors >> 2) & ROOM_LSB, mU = (doors >> 3) & ROOM_LSB;
  uint64_t reach = 1ull << (4 * start);
#pragma unroll 1
  for (int it = 0; it < nrooms; it++) {
    const uint64_t next = reach | ((reach & mR) << 4) | ((reach & mD) << (4 * nc)) | ((reach & mL) >> 4) | ((reach & mU) >> (4 * nc));
    if (next == reach) break;
    reach = next;
  }
  return reach;
}
MG_HD uint64_t rooms_all(int nrooms) { return ROOM_LSB & ((1ull << (4 * nrooms)) - 1ull); }

// ---- general RoomGrid (core/roomgrid.py) for KeyCorridor: 3 columns x up to 3 rows of rooms.  Room bookkeeping is
//      packed: 4 bits per room for the right-door y and the down-door x, one bit per (room, wall) for "has a door or
//      no wall" (Room.doors truthiness), one bit per room for Room.locked. ----
struct RoomGridState {
  int rs, ncols, nrows;
  uint64_t right_y, down_x;     // nibble per room
  uint64_t doors;               // bit room*4 + k, k = right, down, left, up
  uint32_t locked;              // bit per room
  MG_HD int room(int i, int j) const { return j * ncols + i; }
  MG_HD bool has_nb(int i, int j, int k) const { return k == 0 ? i < ncols - 1 : k == 1 ? j < nrows - 1 : k == 2 ? i > 0 : j > 0; }
  MG_HD void door_pos(int i, int j, int k, int& x, int& y) const {     // Room.door_pos[k] (roomgrid.py:158-171)
    if (k == 2) { i -= 1; k = 0; }
    if (k == 3) { j -= 1; k = 1; }
    const int r = room(i, j);
    if (k == 0) { x = i * (rs - 1) + rs - 1; y = (int)((right_y >> (4 * r)) & 15u); }
    else { x = (int)((down_x >> (4 * r)) & 15u); y = j * (rs - 1) + rs - 1; }
  }
  MG_HD void mark(int i, int j, int k) {                               // room.doors[k] and the neighbour's opposite side
    doors |= 1ull << (room(i, j) * 4 + k);
    const int ni = i + (k == 0) - (k == 2), nj = j + (k == 1) - (k == 3);
    doors |= 1ull << (room(ni, nj) * 4 + ((k + 2) & 3));
  }
};

// envs/keycorridor.py:106-133 on RoomGrid._gen_grid / remove_wall / add_door / add_object / place_agent / connect_all
// (roomgrid.py:123-394).  Mission id = COLOR_NAMES index of the ball.
template <class R, class G>
MG_HD void gen_keycorridor(R& rng, G& g, const GenParams& P, GenResult& out) {
  RoomGridState S;
  S.rs = P.room_size; S.ncols = (g.W - 1) / (S.rs - 1); S.nrows = (g.H - 1) / (S.rs - 1);
  S.right_y = 0; S.down_x = 0; S.doors = 0; S.locked = 0;
  const int rs = S.rs, W = g.W, H = g.H;
  if constexpr (G::kWave) {
  g.lattice(rs - 1);
  } else {
    g.clear_empty();
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if ((x % (rs - 1)) == 0 || (y % (rs - 1)) == 0) g.set(x, y, CELL_WALL_GREY);
  }
  for (int j = 0; j < S.nrows; j++)
    for (int i = 0; i < S.ncols; i++) {
      const int tx = i * (rs - 1), ty = j * (rs - 1), r = S.room(i, j);
      if (i < S.ncols - 1) S.right_y |= (uint64_t)rand_int(rng, ty + 1, ty + rs - 1) << (4 * r);
      if (j < S.nrows - 1) S.down_x |= (uint64_t)rand_int(rng, tx + 1, tx + rs - 1) << (4 * r);
    }
  const int mid_x = (S.ncols / 2) * (rs - 1) + rs / 2, mid_y = (S.nrows / 2) * (rs - 1) + rs / 2;   // provisional agent_pos
  for (int j = 1; j < S.nrows; j++) {                       // remove_wall(1, j, 3): the middle column becomes a hallway
    for (int k = 1; k < rs - 1; k++) g.set((rs - 1) + k, j * (rs - 1), CELL_EMPTY);
    S.mark(1, j, 3);
  }
  const int room_idx = rand_int(rng, 0, S.nrows);
  int x, y;
  const uint32_t door_ci = (uint32_t)rand_int(rng, 0, 6);   // add_door(2, room_idx, 2, locked=True)
  S.door_pos(2, room_idx, 2, x, y);
  g.set(x, y, make_cell(T_DOOR_LOCKED, color_from_sorted(door_ci)));
  S.mark(2, room_idx, 2);
  S.locked |= 1u << S.room(2, room_idx);
  const uint32_t ball_ci = (uint32_t)rand_int(rng, 0, 6);   // add_object(2, room_idx, kind="ball")
  if (!place_obj(rng, g, make_cell(T_BALL, color_from_sorted(ball_ci)), 2 * (rs - 1), room_idx * (rs - 1), rs, rs,
                 mid_x, mid_y, true, 1000, x, y)) out.failed = true;
  const int kj = rand_int(rng, 0, S.nrows);                 // add_object(0, rand, "key", door.color)
  if (!place_obj(rng, g, make_cell(T_KEY, color_from_sorted(door_ci)), 0, kj * (rs - 1), rs, rs, mid_x, mid_y, true, 1000, x, y))
    out.failed = true;
  if (!rg_place_agent(rng, g, rs - 1, (S.nrows / 2) * (rs - 1), rs, out)) out.failed = true;
  // connect_all (roomgrid.py:336-394)
  const int start = S.room((int)out.ax / (rs - 1), (int)out.ay / (rs - 1));
  const int nrooms = S.ncols * S.nrows;
  bool connected = rooms_reach(S.doors, start, S.ncols, nrooms) == rooms_all(nrooms);
  for (int itr = 0; !rng.dead(); itr++) {
    if (itr > 5000) { out.failed = true; break; }
    if (connected) break;
    int i = 0, j = 0, k = 0;
    bool picked = false;
    if constexpr (G::kWave) {
      const int n = connect_spec(rng, g, S.ncols, S.nrows, S.doors, S.locked, 5001 - itr, picked, i, j, k);
      if (n > 0) { itr += n - 1; if (!picked) continue; }
    }
    if (!picked) {
      i = rand_int(rng, 0, S.ncols); j = rand_int(rng, 0, S.nrows); k = rand_int(rng, 0, 4);
      if (!S.has_nb(i, j, k) || ((S.doors >> (S.room(i, j) * 4 + k)) & 1ull)) continue;
      const int ni = i + (k == 0) - (k == 2), nj = j + (k == 1) - (k == 3);
      if (((S.locked >> S.room(i, j)) | (S.locked >> S.room(ni, nj))) & 1u) continue;
    }
    const uint32_t ci = (uint32_t)rand_int(rng, 0, 6);
    S.door_pos(i, j, k, x, y);
    g.set(x, y, make_cell(T_DOOR_CLOSED, color_from_sorted(ci)));
    S.mark(i, j, k);
    connected = rooms_reach(S.doors, start, S.ncols, nrooms) == rooms_all(nrooms);
  }
  out.mission = ball_ci;
}

// envs/dynamicobstacles.py:110-134 (P.num_dists = n_obstacles after the constructor's clamp; grids up to 16x16)
template <class R>
MG_D void gen_dynobs(R& rng, GridRef& g, const GenParams& P, GenResult& out) {
  g.clear_with_walls();
  g.set(g.W - 2, g.H - 2, CELL_GOAL);
  if (P.start_x >= 0) { out.ax = P.start_x; out.ay = P.start_y; out.dir = P.start_dir; }
  else if (!place_agent(rng, g, 0, 0, g.W, g.H, -1, out)) out.failed = true;
  uint64_t obst = 0;
  const int n = min(P.num_dists, 8);
  for (int i = 0; i < n; i++) {
    int x = 0, y = 0;
    if (!place_obj(rng, g, CELL_BALL_BLUE, 0, 0, g.W, g.H, (int)out.ax, (int)out.ay, false, 100, x, y)) out.failed = true;
    obst |= (uint64_t)(y * g.W + x) << (8 * i);
  }
  out.aux = obst;
  out.mission = 0;
}

// envs/redbluedoors.py:78-102 (size = H, width = 2 * size)
template <class R, class G>
MG_HD void gen_redbluedoors(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int S = g.H, W = g.W;
  if constexpr (G::kWave) {
  MG_WAVE_LDS_SYNC();
  for (int y = 0; y < S; y++)
    if (g.lane < W) {           // wall_rect(0, 0, 2S, S) and wall_rect(S/2, 0, S, S)
      const bool wall = y == 0 || y == S - 1 || g.lane == 0 || g.lane == W - 1 || g.lane == S / 2 || g.lane == S / 2 + S - 1;
      g.p[y * W + g.lane] = (uint8_t)(wall ? CELL_WALL_GREY : CELL_EMPTY);
    }
  MG_WAVE_LDS_SYNC();
  } else {
    g.clear_empty();
    for (int y = 0; y < S; y++) for (int x = 0; x < W; x++)
      if (y == 0 || y == S - 1 || x == 0 || x == W - 1 || x == S / 2 || x == S / 2 + S - 1) g.set(x, y, CELL_WALL_GREY);
  }
  if (!place_agent(rng, g, S / 2, 0, S, S, -1, out)) out.failed = true;
  g.set(S / 2, rand_int(rng, 1, S - 1), make_cell(T_DOOR_CLOSED, C_RED));
  g.set(S / 2 + S - 1, rand_int(rng, 1, S - 1), make_cell(T_DOOR_CLOSED, C_BLUE));
  out.mission = 0;
}

// envs/memory.py:92-149
template <class R, class G>
MG_HD void gen_memory(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H, mid = H / 2;
  g.clear_with_walls();
  const int upper = mid - 2, lower = mid + 2;
  const int he = P.random_length ? rand_int(rng, 4, W - 2) : W - 3;
  for (int i = 1; i < 5; i++) { g.set(i, upper, CELL_WALL_GREY); g.set(i, lower, CELL_WALL_GREY); }
  g.set(4, upper + 1, CELL_WALL_GREY); g.set(4, lower - 1, CELL_WALL_GREY);
  for (int i = 5; i < he; i++) { g.set(i, upper + 1, CELL_WALL_GREY); g.set(i, lower - 1, CELL_WALL_GREY); }
  for (int j = 0; j < H; j++) {
    if (j != mid) g.set(he, j, CELL_WALL_GREY);
    g.set(he + 2, j, CELL_WALL_GREY);
  }
  out.ax = (uint32_t)rand_int(rng, 1, he + 1); out.ay = (uint32_t)mid; out.dir = 0;
  const uint32_t start_ball = (uint32_t)rand_int(rng, 0, 2);                   // _rand_elem([Key, Ball])
  g.set(1, mid - 1, make_cell(start_ball ? (uint32_t)T_BALL : (uint32_t)T_KEY, C_GREEN));
  const uint32_t top_ball = rand_int(rng, 0, 2) == 0;                          // _rand_elem([[Ball, Key], [Key, Ball]])
  g.set(he + 1, mid - 2, make_cell(top_ball ? (uint32_t)T_BALL : (uint32_t)T_KEY, C_GREEN));
  g.set(he + 1, mid + 2, make_cell(top_ball ? (uint32_t)T_KEY : (uint32_t)T_BALL, C_GREEN));
  out.mission = 0;
}

// BabyAI levels with a single ActionInstr on small RoomGrids and no check_objs_reachable in their gen_mission:
// PickupDist / PickupDistDebug (envs/babyai/pickup.py:276-290: 7x7 room, add_distractors(5, all_unique) then
// place_agent(0, 0)), OneRoomS8..S20 (other.py:329-332: add_object(0, 0, "ball") then place_agent()).
// Mission id of "pick up " + ObjDesc.surface (verifier.py:73-103): article ("the" 0 | "a" 1) * 28 +
// (no colour 0 | COLOR_NAMES index + 1) * 4 + ("object" 0 | key 1 | ball 2 | box 3).
enum : int { KIND_PICKUPDIST = 24, KIND_ONEROOM = 25, KIND_OPENREDDOOR = 26, KIND_PICKUPDIST_DEBUG = 27 };
template <class R, class G>
MG_HD void gen_pickup_level(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H, mid = W / 2;      // 1x1 RoomGrid: the provisional agent_pos reject_next_to sees (roomgrid.py:174-179)
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();                           // a RecursionError regenerates from the current stream position
    g.clear_with_walls();
    uint32_t dcol = 0, dtyp = 0, used = 0;      // nibble k = colour / type index of object k; bit colour * 3 + type
    int n = 0, x, y;
    bool ok = true;
    if (P.kind == KIND_ONEROOM) {
      dcol = (uint32_t)rand_int(rng, 0, 6); dtyp = 1u; n = 1;       // add_object(kind="ball"): _rand_color(), then place_in_room
      ok = place_obj(rng, g, make_cell(T_BALL, color_from_sorted(dcol)), 0, 0, W, H, mid, mid, true, 1000, x, y);
    } else {
      while (n < 5 && ok && !rng.dead()) {                           // add_distractors (roomgrid.py:396-438)
        const uint32_t ci = (uint32_t)rand_int(rng, 0, 6), ti = (uint32_t)rand_int(rng, 0, 3), id = ci * 3u + ti;
        if ((used >> id) & 1u) continue;
        ok = place_obj(rng, g, make_cell((uint32_t)T_KEY + ti, color_from_sorted(ci)), 0, 0, W, H, mid, mid, true, 1000, x, y);
        used |= 1u << id; dcol |= ci << (4 * n); dtyp |= ti << (4 * n); n++;
      }
    }
    if (!ok) continue;
    if (!rg_place_agent(rng, g, 0, 0, W, out)) continue;
    uint32_t ci = 0, ti = 2;                                          // OneRoom: ObjDesc("ball")
    if (P.kind != KIND_ONEROOM) {
      const int k = rand_int(rng, 0, n);
      const int sel = rand_int(rng, 0, 3);                            // _rand_elem(["type", "color", "both"])
      ci = sel == 0 ? 0u : ((dcol >> (4 * k)) & 15u) + 1u;
      ti = sel == 1 ? 0u : ((dtyp >> (4 * k)) & 15u) + 1u;
    }
    uint32_t matches = 0;
    for (int k = 0; k < n; k++)
      matches += (ci == 0u || ((dcol >> (4 * k)) & 15u) == ci - 1u) && (ti == 0u || ((dtyp >> (4 * k)) & 15u) == ti - 1u) ? 1u : 0u;
    if (ti == 0u && ci == 3u) matches += 2u;    // "grey" without a type also matches every wall (verifier.py:139-146)
    out.mission = (matches > 1u ? 28u : 0u) + ci * 4u + ti;
    (void)H;
    return;
  }
  out.failed = true;
}
// envs/babyai/other.py:169-177 (FindObjS5/S6/S7) on RoomGrid._gen_grid / add_object / place_agent / connect_all
// (core/roomgrid.py:123-394) for 3 x 3 rooms: a random object in a random room, the agent in the middle room, doors added
// at random until every room is reachable; PickupInstr(ObjDesc(obj.type)) -> mission "pick up the <type>".
// Room bookkeeping: the door offsets inside the room (1 .. rs-2: a nibble each; S7's absolute coordinates reach 17) and
// one bit per (room, wall) for Room.doors.
template <class R, class G>
MG_HD void gen_findobj(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int rs = P.room_size, W = g.W, H = g.H, st = rs - 1;
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();                           // a RecursionError regenerates from the current stream position
    if constexpr (G::kWave) {
    g.lattice(st);
    } else {
      g.clear_empty();
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if ((x % st) == 0 || (y % st) == 0) g.set(x, y, CELL_WALL_GREY);
    }
    uint64_t right_y = 0, down_x = 0, doors = 0;
#pragma unroll 1
    for (int j = 0; j < 3; j++)
#pragma unroll 1
      for (int i = 0; i < 3; i++) {
        const int r = j * 3 + i, tx = i * st, ty = j * st;
        if (i < 2) right_y |= (uint64_t)(rand_int(rng, ty + 1, ty + rs - 1) - ty) << (4 * r);
        if (j < 2) down_x |= (uint64_t)(rand_int(rng, tx + 1, tx + rs - 1) - tx) << (4 * r);
      }
    const int mid = st + rs / 2;                // provisional agent_pos = middle of the middle room (roomgrid.py:174-179)
    const int oi = rand_int(rng, 0, 3), oj = rand_int(rng, 0, 3);                 // get_room(i, j): i is the column
    const uint32_t ti = (uint32_t)rand_int(rng, 0, 3), ci = (uint32_t)rand_int(rng, 0, 6);      // add_object: kind, then colour
    int x, y;
    if (!place_obj(rng, g, make_cell((uint32_t)T_KEY + ti, color_from_sorted(ci)), oi * st, oj * st, rs, rs, mid, mid, true, 1000, x, y)) continue;
    if (!rg_place_agent(rng, g, st, st, rs, out)) continue;
    // connect_all (roomgrid.py:336-394)
    const int start = ((int)out.ay / st) * 3 + (int)out.ax / st;
    bool fail = false;
    bool connected = rooms_reach(doors, start, 3, 9) == rooms_all(9);
    for (int itr = 0; !rng.dead(); itr++) {
      if (itr > 5000) { fail = true; break; }
      if (connected) break;
      int i = 0, j = 0, k = 0;
      bool picked = false;
      if constexpr (G::kWave) {
        const int n = connect_spec(rng, g, 3, 3, doors, 0u, 5001 - itr, picked, i, j, k);
        if (n > 0) { itr += n - 1; if (!picked) continue; }
      }
      if (!picked) {
        i = rand_int(rng, 0, 3); j = rand_int(rng, 0, 3); k = rand_int(rng, 0, 4);
        const bool has_nb = k == 0 ? i < 2 : k == 1 ? j < 2 : k == 2 ? i > 0 : j > 0;
        if (!has_nb || ((doors >> ((j * 3 + i) * 4 + k)) & 1ull)) continue;
      }
      const int r = j * 3 + i;
      const uint32_t dc = (uint32_t)rand_int(rng, 0, 6);
      // Room.door_pos[k] (roomgrid.py:158-171): the left / up door is the neighbour's right / down door
      const int ri = k == 2 ? i - 1 : i, rj = k == 3 ? j - 1 : j, rr = rj * 3 + ri;
      const bool vertical_wall = k == 0 || k == 2;
      const int dx = vertical_wall ? ri * st + st : ri * st + (int)((down_x >> (4 * rr)) & 15u);
      const int dy = vertical_wall ? rj * st + (int)((right_y >> (4 * rr)) & 15u) : rj * st + st;
      g.set(dx, dy, make_cell(T_DOOR_CLOSED, color_from_sorted(dc)));
      const int nr = r + (k == 0 ? 1 : k == 1 ? 3 : k == 2 ? -1 : -3);
      doors |= (1ull << (r * 4 + k)) | (1ull << (nr * 4 + ((k + 2) & 3)));
      connected = rooms_reach(doors, start, 3, 9) == rooms_all(9);
    }
    if (fail) continue;
    out.mission = ti + 1u;                      // "pick up the key / ball / box"
    return;
  }
  out.failed = true;
}
// envs/babyai/unlock.py:167-174 (UnlockLocal / UnlockLocalDist: 3 x 3 rooms; a locked door on a random wall of the middle
// room, its key and P.num_dists (0 | 3) distractors in that room, the agent too; OpenInstr(ObjDesc("door")) -> "open the door")
template <class R, class G>
MG_HD void gen_unlocklocal(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int rs = P.room_size, W = g.W, H = g.H, st = rs - 1;
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();
    if constexpr (G::kWave) {
    g.lattice(st);
    } else {
      g.clear_empty();
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if ((x % st) == 0 || (y % st) == 0) g.set(x, y, CELL_WALL_GREY);
    }
    uint64_t right_y = 0, down_x = 0;           // RoomGrid._gen_grid: one door offset per room and shared wall (roomgrid.py:158-171)
#pragma unroll 1
    for (int j = 0; j < 3; j++)
#pragma unroll 1
      for (int i = 0; i < 3; i++) {
        const int r = j * 3 + i, tx = i * st, ty = j * st;
        if (i < 2) right_y |= (uint64_t)(rand_int(rng, ty + 1, ty + rs - 1) - ty) << (4 * r);
        if (j < 2) down_x |= (uint64_t)(rand_int(rng, tx + 1, tx + rs - 1) - tx) << (4 * r);
      }
    const int mid = st + rs / 2;
    // add_door(1, 1, locked=True): the first door_idx drawn is valid (four neighbours, no door yet), then the colour
    const int k = rand_int(rng, 0, 4);
    const uint32_t dc = (uint32_t)rand_int(rng, 0, 6);
    const int ri = k == 2 ? 0 : 1, rj = k == 3 ? 0 : 1, rr = rj * 3 + ri;
    const bool vertical_wall = k == 0 || k == 2;
    const int dx = vertical_wall ? ri * st + st : ri * st + (int)((down_x >> (4 * rr)) & 15u);
    const int dy = vertical_wall ? rj * st + (int)((right_y >> (4 * rr)) & 15u) : rj * st + st;
    g.set(dx, dy, make_cell(T_DOOR_LOCKED, color_from_sorted(dc)));
    int x, y;
    bool ok = place_obj(rng, g, make_cell(T_KEY, color_from_sorted(dc)), st, st, rs, rs, mid, mid, true, 1000, x, y);
    uint32_t used = 1u << (dc * 3u);            // add_distractors: (type, colour) pairs already in some room.objs
    for (int n = 0; n < P.num_dists && ok && !rng.dead();) {
      const uint32_t ci = (uint32_t)rand_int(rng, 0, 6), ti = (uint32_t)rand_int(rng, 0, 3), id = ci * 3u + ti;
      if ((used >> id) & 1u) continue;
      ok = place_obj(rng, g, make_cell((uint32_t)T_KEY + ti, color_from_sorted(ci)), st, st, rs, rs, mid, mid, true, 1000, x, y);
      used |= 1u << id; n++;
    }
    if (!ok) continue;
    if (!rg_place_agent(rng, g, st, st, rs, out)) continue;
    out.mission = 0;
    return;
  }
  out.failed = true;
}
// envs/obstructedmaze.py:111-270 and envs/obstructedmaze_v1.py:37-100 on RoomGrid._gen_grid / add_door / add_object /
// place_in_room / place_agent (no connect_all).  room_size 6; P.num_crossings = flags (1 key_in_box, 2 blocked, 4 the v1
// class, 8 the 1 x 2 class ObstructedMaze_1Dlhb); P.num_dists = num_quarters; P.start_x / start_y = agent_room.  The ball
// to find is COLOR_NAMES[0] (blue), blocking balls COLOR_NAMES[1] (green), boxes COLOR_NAMES[2] (grey).
template <class R, class G>
MG_HD void gen_obstructedmaze(R& rng, G& g, const GenParams& P, GenResult& out) {
  RoomGridState S;
  S.rs = P.room_size; S.ncols = (g.W - 1) / (S.rs - 1); S.nrows = (g.H - 1) / (S.rs - 1);
  S.right_y = 0; S.down_x = 0; S.doors = 0; S.locked = 0;
  const int rs = S.rs, st = rs - 1, W = g.W, H = g.H;
  const int flags = P.num_crossings;
  const bool key_in_box = flags & 1, blocked = (flags >> 1) & 1, v1 = (flags >> 2) & 1, one_d = (flags >> 3) & 1;
  if constexpr (G::kWave) {
  g.lattice(st);
  } else {
    g.clear_empty();
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if ((x % st) == 0 || (y % st) == 0) g.set(x, y, CELL_WALL_GREY);
  }
#pragma unroll 1
  for (int j = 0; j < S.nrows; j++)
#pragma unroll 1
    for (int i = 0; i < S.ncols; i++) {
      const int tx = i * st, ty = j * st, r = S.room(i, j);
      if (i < S.ncols - 1) S.right_y |= (uint64_t)rand_int(rng, ty + 1, ty + rs - 1) << (4 * r);
      if (j < S.nrows - 1) S.down_x |= (uint64_t)rand_int(rng, tx + 1, tx + rs - 1) << (4 * r);
    }
  const int mid_x = (S.ncols / 2) * st + rs / 2, mid_y = (S.nrows / 2) * st + rs / 2;       // provisional agent_pos
  // door_colors = _rand_subset(COLOR_NAMES, 6) (minigrid_env.py:277-292): six draws without replacement, 4 bits each
  uint32_t door_colors = 0, avail = 0x543210u;
#pragma unroll 1
  for (int n = 0, na = 6; n < 6; n++, na--) {
    const int k = rand_int(rng, 0, na);
    door_colors |= ((avail >> (4 * k)) & 15u) << (4 * n);
    const uint32_t lowmask = (1u << (4 * k)) - 1u;
    avail = (avail & lowmask) | ((avail >> 4) & ~lowmask);
  }
  bool ok = true;
  auto add_key = [&](int i, int j, uint32_t ci) {                          // place_in_room(i, j, key or box(key))
    const uint32_t col = color_from_sorted(ci);
    const uint32_t cell = key_in_box ? make_cell(T_BOX_KEY, col) : make_cell(T_KEY, col);
    int x, y;
    if (!place_obj(rng, g, cell, i * st, j * st, rs, rs, mid_x, mid_y, true, 1000, x, y)) ok = false;
  };
  auto add_door = [&](int i, int j, int k, uint32_t ci, bool locked, bool with_key) {
    int x, y;
    S.door_pos(i, j, k, x, y);
    g.set(x, y, make_cell(locked ? (uint32_t)T_DOOR_LOCKED : (uint32_t)T_DOOR_CLOSED, color_from_sorted(ci)));
    S.mark(i, j, k);
    if (locked && blocked) g.set(x - dir_dx((uint32_t)k), y - dir_dy((uint32_t)k), make_cell(T_BALL, color_from_sorted(1u)));
    if (locked && with_key) add_key(i, j, ci);
  };
  int x, y;
  if (one_d) {
    add_door(0, 0, 0, door_colors & 15u, true, true);
    if (!place_obj(rng, g, make_cell(T_BALL, color_from_sorted(0u)), st, 0, rs, rs, mid_x, mid_y, true, 1000, x, y)) ok = false;
    if (!rg_place_agent(rng, g, 0, 0, rs, out)) ok = false;
  } else {
    const int nq = P.num_dists;
#pragma unroll 1
    for (int i = 0; i < nq; i++) {
      const int si = i == 0 ? 2 : (i == 2 ? 0 : 1), sj = i == 1 ? 2 : (i == 3 ? 0 : 1);      // side_rooms (2,1) (1,2) (0,1) (1,0)
      add_door(1, 1, i, (door_colors >> (4 * i)) & 15u, false, false);
#pragma unroll 1
      for (int k = -1; k <= 1; k += 2)
        add_door(si, sj, (i + k + 4) % 4, (door_colors >> (4 * ((i + k + 6) % 6))) & 15u, true, !v1);
      if (v1)                                                                 // v1: keys after both doors and their blocking balls
        for (int k = -1; k <= 1; k += 2) add_key(si, sj, (door_colors >> (4 * ((i + k + 6) % 6))) & 15u);
    }
    const int c = rand_int(rng, 0, nq);                                       // corners (2,0) (2,2) (0,2) (0,0)
    const int ci = c < 2 ? 2 : 0, cj = (c == 1 || c == 2) ? 2 : 0;
    if (!place_obj(rng, g, make_cell(T_BALL, color_from_sorted(0u)), ci * st, cj * st, rs, rs, mid_x, mid_y, true, 1000, x, y)) ok = false;
    if (!rg_place_agent(rng, g, P.start_x * st, P.start_y * st, rs, out)) ok = false;
  }
  if (!ok) out.failed = true;
  out.mission = 0;
}

// envs/babyai/open.py:143-146 (OpenRedDoor: 1 x 2 rooms of size 5; add_door(0, 0, 0, "red", locked=False); place_agent(0, 0))
template <class R, class G>
MG_HD void gen_openreddoor(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int rs = P.room_size, W = g.W, H = g.H;
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();
    if constexpr (G::kWave) {
    MG_WAVE_LDS_SYNC();
    for (int y = 0; y < H; y++)
      if (g.lane < W) {
        const bool wall = y == 0 || y == H - 1 || (g.lane % (rs - 1)) == 0;
        g.p[y * W + g.lane] = (uint8_t)(wall ? CELL_WALL_GREY : CELL_EMPTY);
      }
    MG_WAVE_LDS_SYNC();
    } else {
      g.clear_empty();
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if (y == 0 || y == H - 1 || (x % (rs - 1)) == 0) g.set(x, y, CELL_WALL_GREY);
    }
    const int door_y = rand_int(rng, 1, rs - 1);                      // room (0,0).door_pos[0] (roomgrid.py:158-163)
    g.set(rs - 1, door_y, make_cell(T_DOOR_CLOSED, C_RED));
    if (!rg_place_agent(rng, g, 0, 0, rs, out)) continue;
    out.mission = 0;
    return;
  }
  out.failed = true;
}


// envs/lockedroom.py:104-176 (19x19: six rooms off a central hallway, one locked with the goal inside, the key of its
// colour in another room).  Mission id = COLOR_NAMES index of the locked room * 6 + COLOR_NAMES index of the key room.
template <class R, class G>
MG_HD void gen_lockedroom(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H;
  g.clear_with_walls();
  const int lw = W / 2 - 2, rw = W / 2 + 2, third = H / 3;
  for (int j = 0; j < H; j++) { g.set(lw, j, CELL_WALL_GREY); g.set(rw, j, CELL_WALL_GREY); }
#pragma unroll 1
  for (int n = 0; n < 3; n++) {
    const int j = n * third;
    for (int i = 0; i < lw; i++) g.set(i, j, CELL_WALL_GREY);
    for (int i = rw; i < W; i++) g.set(i, j, CELL_WALL_GREY);
  }
  // room r = 2n + side: top = (side ? rw : 0, n * third), size (lw + 1, third + 1), door (side ? rw : lw, n * third + 3)
  const int room_w = lw + 1, room_h = third + 1;
  const int locked = rand_int(rng, 0, 6);
  {
    const int tx = (locked & 1) ? rw : 0, ty = (locked >> 1) * third;
    const int gx = rand_int(rng, tx + 1, tx + room_w - 1);          // LockedRoom.rand_pos = _rand_pos: x, then y
    const int gy = rand_int(rng, ty + 1, ty + room_h - 1);
    g.set(gx, gy, CELL_GOAL);
  }
  uint32_t avail = 0x543210u, colors = 0;       // sorted(colors) still unassigned, one nibble each; nibble r = colour of room r
#pragma unroll 1
  for (int r = 0, na = 6; r < 6; r++, na--) {
    const int k = rand_int(rng, 0, na);
    const uint32_t ci = (avail >> (4 * k)) & 15u;
    const uint32_t lo = avail & ((1u << (4 * k)) - 1u);
    avail = lo | ((avail >> (4 * (k + 1))) << (4 * k));
    colors |= ci << (4 * r);
    g.set((r & 1) ? rw : lw, (r >> 1) * third + 3, make_cell(r == locked ? (uint32_t)T_DOOR_LOCKED : (uint32_t)T_DOOR_CLOSED, color_from_sorted(ci)));
  }
  int key_room = locked;
  while (key_room == locked && !rng.dead()) key_room = rand_int(rng, 0, 6);
  const uint32_t lc = (colors >> (4 * locked)) & 15u, kc = (colors >> (4 * key_room)) & 15u;
  {
    const int tx = (key_room & 1) ? rw : 0, ty = (key_room >> 1) * third;
    const int kx = rand_int(rng, tx + 1, tx + room_w - 1);
    const int ky = rand_int(rng, ty + 1, ty + room_h - 1);
    g.set(kx, ky, make_cell(T_KEY, color_from_sorted(lc)));
  }
  if (!place_agent(rng, g, lw, 0, rw - lw, H, -1, out)) out.failed = true;
  out.mission = lc * 6u + kc;
}

// envs/playground.py:31-91: 3x3 rooms, one door per shared wall, 12 random objects; no goal, one (empty) mission
template <class R, class G>
MG_HD void gen_playground(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W, H = g.H;
  g.clear_with_walls();
  const int room_w = W / 3, room_h = H / 3;
#pragma unroll 1
  for (int j = 0; j < 3; j++) {
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
      const int xL = i * room_w, yT = j * room_h, xR = xL + room_w, yB = yT + room_h;
      if (i + 1 < 3) {
        for (int k = 0; k < room_h; k++) g.set(xR, yT + k, CELL_WALL_GREY);
        const int py = rand_int(rng, yT + 1, yB - 1);
        const uint32_t c = (uint32_t)rand_int(rng, 0, 6);
        g.set(xR, py, make_cell(T_DOOR_CLOSED, color_from_sorted(c)));
      }
      if (j + 1 < 3) {
        for (int k = 0; k < room_w; k++) g.set(xL + k, yB, CELL_WALL_GREY);
        const int px = rand_int(rng, xL + 1, xR - 1);
        const uint32_t c = (uint32_t)rand_int(rng, 0, 6);
        g.set(px, yB, make_cell(T_DOOR_CLOSED, color_from_sorted(c)));
      }
    }
  }
  if (!place_agent(rng, g, 0, 0, W, H, -1, out)) out.failed = true;
  int x, y;
#pragma unroll 1
  for (int k = 0; k < 12; k++) {
    const uint32_t t = (uint32_t)rand_int(rng, 0, 3), c = (uint32_t)rand_int(rng, 0, 6);
    if (!place_obj(rng, g, make_cell((uint32_t)T_KEY + t, color_from_sorted(c)), 0, 0, W, H, (int)out.ax, (int)out.ay, false, -1, x, y)) out.failed = true;
  }
  out.mission = 0;
}

// envs/multiroom.py:118-300 (P.num_crossings = minNumRooms, P.num_dists = maxNumRooms <= 6, P.room_size = maxRoomSize).
// _placeRoom's recursion is a chain: a placed room makes up to 8 attempts at the next one and stops at the first that
// fits (a child that was appended always returns True, :293-296); rooms are never removed; the outer loop keeps the
// longest chain until one has numRooms rooms.  A room is one word: tx | ty << 5 | sx << 10 | sy << 14 | ex << 18 |
// ey << 23.  The chain being built lives at the start of the (not yet drawn) grid; what has to survive a restart from
// a checkpoint -- numRooms, the best chain -- in the wave's scratch words: an N6 map can take > 1000 draws, more than one buffer.
MG_HD uint32_t mr_pack(int tx, int ty, int sx, int sy, int ex, int ey) {
  return (uint32_t)tx | ((uint32_t)ty << 5) | ((uint32_t)sx << 10) | ((uint32_t)sy << 14) | ((uint32_t)ex << 18) | ((uint32_t)ey << 23);
}
// (templated on the grid type like the other generators: one lane's form keeps the state words in registers -- a lane's stream never runs dry,
// there is no restart -- and is pinned on the CPU by tests/test_generators_cpu.py)
template <class G> MG_HD uint32_t mr_word(uint32_t v) { if constexpr (G::kWave) return uni32(v); else return v; }
template <bool WAVE> struct MrState {
  uint32_t* p;
  MG_HD MrState(uint8_t* grid, int off) : p((uint32_t*)(grid + off)) {}
};
template <> struct MrState<false> {
  uint32_t w[8]; uint32_t* p;
  MG_HD MrState(uint8_t*, int) : p(w) {}
};
template <class R, class G>
MG_HD bool mr_try_room(R& rng, const G& g, const GenParams& P, uint32_t* cur, int& n, int wall, int ex, int ey) {
  const int sx = rand_int(rng, 4, P.room_size + 1), sy = rand_int(rng, 4, P.room_size + 1);
  int tx, ty;
  if (n == 0) { tx = ex; ty = ey; }
  else if (wall == 0) { tx = ex - sx + 1; ty = rand_int(rng, ey - sy + 2, ey); }
  else if (wall == 1) { tx = rand_int(rng, ex - sx + 2, ex); ty = ey - sy + 1; }
  else if (wall == 2) { tx = ex; ty = rand_int(rng, ey - sy + 2, ey); }
  else { tx = rand_int(rng, ex - sx + 2, ex); ty = ey; }
  if (tx < 0 || ty < 0) return false;
  if (tx + sx > g.W || ty + sy >= g.H) return false;
#pragma unroll 1
  for (int k = 0; k + 1 < n; k++) {                    // roomList[:-1]
    const uint32_t r = mr_word<G>(cur[k]);
    const int rtx = (int)(r & 31u), rty = (int)((r >> 5) & 31u), rsx = (int)((r >> 10) & 15u), rsy = (int)((r >> 14) & 15u);
    const bool non_overlap = tx + sx < rtx || rtx + rsx <= tx || ty + sy < rty || rty + rsy <= ty;
    if (!non_overlap) return false;
  }
  cur[n++] = mr_pack(tx, ty, sx, sy, ex, ey);
  return true;
}
// _placeRoom's `for i in range(0, 8)` loop (multiroom.py:247-281) of the newest room, speculatively (round 6): until a child room is accepted nothing the
// loop looks at changes, and every try draws the same number of values whatever its outcome (exit wall, door offset, [sizeX, sizeY], top offset: the checks of
// :223-241 come after the draws), so lane t evaluates try t from draws wpos + per * t .. against the rooms placed so far; the first lane whose room fits is
// the try the reference accepts, the lanes before it are its rejected tries.  Returns the tries consumed (with their draws); `acc`: the last of them placed
// `room` behind entry wall `entry`.  0 = nothing could be evaluated (no buffered draws left, or try 0 holds a draw numpy's Lemire step might reject): the
// caller runs one try the scalar way.  The scalar chain search was 65 % of a MultiRoom-N6 episode (profiles/r6/refill_attribution_multiroom_*.txt).
template <class R, class G>
MG_D int mr_spec(R& rng, const G& g, const GenParams& P, const uint32_t* cur, int n, int wall, int max_tries, bool& acc, uint32_t& room, int& entry) {
  acc = false;
  if (rng.dead()) return 0;
  const uint32_t par = uni32(cur[n - 1]);
  const int ptx = (int)(par & 31u), pty = (int)((par >> 5) & 31u), psx = (int)((par >> 10) & 15u), psy = (int)((par >> 14) & 15u);
  const uint32_t nsz = P.room_size > 4 ? 2u : 0u;                   // _rand_int(4, maxSz + 1) over one value draws nothing
  const uint32_t per = 3u + nsz;
  const uint32_t p0 = rng.wpos + per * (uint32_t)g.lane;
  const bool have = p0 + per <= rng.window() && g.lane < max_tries;
  bool s0 = true, s1 = true, s2 = true, s3 = true, s4 = true;
  const int k = (int)peek_bounded(rng, p0, 3u, s0);                 // _rand_elem(sorted({0,1,2,3} - {entryDoorWall}))
  const int exit_wall = k >= wall ? k + 1 : k, next_entry = (exit_wall + 2) & 3;
  const bool ew_x = (exit_wall & 1) == 0;                            // exit on the right / left wall: the door's y is drawn
  const int d = 1 + (int)peek_bounded(rng, p0 + 1u, (uint32_t)((ew_x ? psy : psx) - 2), s1);
  const int ex = exit_wall == 0 ? ptx + psx - 1 : exit_wall == 2 ? ptx : ptx + d;
  const int ey = exit_wall == 1 ? pty + psy - 1 : exit_wall == 3 ? pty : pty + d;
  int sx = 4, sy = 4;
  if (nsz) { sx += (int)peek_bounded(rng, p0 + 2u, (uint32_t)(P.room_size - 3), s2); sy += (int)peek_bounded(rng, p0 + 3u, (uint32_t)(P.room_size - 3), s3); }
  const bool ne_x = (next_entry & 1) == 0;                           // entry on the child's right / left wall: its topY is drawn
  const int t = (int)peek_bounded(rng, p0 + 2u + nsz, (uint32_t)((ne_x ? sy : sx) - 2), s4);
  const int tx = next_entry == 0 ? ex - sx + 1 : next_entry == 2 ? ex : ex - sx + 2 + t;
  const int ty = next_entry == 1 ? ey - sy + 1 : next_entry == 3 ? ey : ey - sy + 2 + t;
  const unsigned long long unsafe = __ballot(have && !(s0 && s1 && s2 && s3 && s4));
  const int nhave = __popcll(__ballot(have));                       // (the lanes that have their draws are a prefix)
  const int nt = unsafe ? min(nhave, __ffsll((long long)unsafe) - 1) : nhave;
  if (nt == 0) return 0;
  bool ok = tx >= 0 && ty >= 0 && tx + sx <= g.W && ty + sy < g.H;
#pragma unroll 1
  for (int q = 0; q + 1 < n; q++) {                                  // roomList[:-1]
    const uint32_t r = uni32(cur[q]);
    const int rtx = (int)(r & 31u), rty = (int)((r >> 5) & 31u), rsx = (int)((r >> 10) & 15u), rsy = (int)((r >> 14) & 15u);
    ok = ok && (tx + sx < rtx || rtx + rsx <= tx || ty + sy < rty || rty + rsy <= ty);
  }
  const unsigned long long okm = __ballot(g.lane < nt && ok);
  if (okm) {
    const int a = __ffsll((long long)okm) - 1;
    rng.wpos += per * (uint32_t)(a + 1);
    room = lane32(mr_pack(tx, ty, sx, sy, ex, ey), (uint32_t)a);
    entry = (int)lane32((uint32_t)next_entry, (uint32_t)a);
    acc = true;
    return a + 1;
  }
  rng.wpos += per * (uint32_t)nt;
  return nt;
}
// ---- MultiRoom's chain search for ONE LANE per episode (k_refill_lane_packed / k_generate_lane), round 6 ----
// The c (<= 5) next 32-bit draws of a lane's stream at once.  numpy hands out the halves of one 64-bit PCG64 output one after the other (mg_rng.h
// Pcg64Stream::next32), so the lanes of a wave disagree about WHICH of two consecutive draws needs the 128-bit LCG step and a wave walking through
// next32() five times executes five steps, each for half of its lanes.  Here every lane takes ceil((c - has_uint32) / 2) steps -- the first two under no
// lane condition at all for c = 4 or 5 -- and picks its words out of [cached half] o1.lo o1.hi o2.lo o2.hi o3.lo o3.hi; what is left over is the new
// cached half: the stream position afterwards is exactly that of c calls of next32().
template <class R> MG_HD void next32_block(R& r, int c, uint32_t w[5]) {
#pragma unroll
  for (int k = 0; k < 5; k++) w[k] = k < c ? r.next32() : 0u;
}
MG_HD void next32_block(Pcg64Stream& r, int c, uint32_t w[5]) {
  const uint32_t h = r.has32;
  const int need = (c - (int)h + 1) >> 1;
  uint64_t o1 = 0, o2 = 0, o3 = 0;
  if (need >= 1) o1 = r.next64();
  if (need >= 2) o2 = r.next64();
  if (need >= 3) o3 = r.next64();
  const uint32_t L[8] = { r.cache32, (uint32_t)o1, (uint32_t)(o1 >> 32), (uint32_t)o2, (uint32_t)(o2 >> 32), (uint32_t)o3, (uint32_t)(o3 >> 32), 0u };
#pragma unroll
  for (int k = 0; k < 5; k++) w[k] = h ? L[k] : L[k + 1];
  // the word after the last one handed out, if a step produced it: index c (cached half in front) or c + 1
  const int j = h ? c : c + 1;
  const uint32_t next = j == 1 ? L[1] : j == 2 ? L[2] : j == 3 ? L[3] : j == 4 ? L[4] : j == 5 ? L[5] : L[6];
  const bool left = (int)h + 2 * need - c == 1;
  r.has32 = left ? 1u : 0u;
  r.cache32 = left ? next : r.cache32;
}
// _gen_grid's `while len(roomList) < numRooms` (multiroom.py:123-141) with _placeRoom's recursion (:193-283) as ONE flat loop: an iteration is one
// room placement try -- the first room of a new attempt (draws: entry x, entry y, [sizeX, sizeY]) or one of the eight tries at the next room (draws:
// exit wall, door offset, [sizeX, sizeY], top offset) -- so the 64 lanes of a wave, each on its own episode, execute the same instructions whatever
// attempt / room / try they are at, and a wave runs max-over-lanes(tries) iterations instead of the sum of the maxima of three nested loops.
// The try is evaluated from the raw words with Lemire's multiply; a draw numpy's rejection step could apply to (low product word below the range:
// probability ~ range / 2^32) sends that lane through the same evaluation on rand_int, from the stream position saved before the block.
// `bw` = the best chain (what gen_multiroom draws), `cur` = the chain under construction at the start of the lane's (not yet drawn) grid.
template <class R>
MG_HD void mr_search_lane(R& rng, const LaneGrid& g, const GenParams& P, int num_rooms, uint32_t bw[6], int& nbest) {
  const int W = g.W, H = g.H;
  uint32_t* cur = (uint32_t*)g.p;
  const int nsz = P.room_size > 4 ? 2 : 0;                 // _rand_int(4, maxSz + 1) over one value draws nothing
  const uint32_t rsz = (uint32_t)(P.room_size - 3);
  int n = 0, wall = 2, i = 0;
  uint32_t parent = 0;
  nbest = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) bw[k] = 0u;
  while (nbest < num_rooms) {
    const bool first = n == 0;
    const int ptx = (int)(parent & 31u), pty = (int)((parent >> 5) & 31u), psx = (int)((parent >> 10) & 15u), psy = (int)((parent >> 14) & 15u);
    bool ok = false; uint32_t room = 0; int entry = 2;
    // one try, its draws taken from `draw(k, r)` = the try's k-th draw over r >= 2 values
    auto eval = [&](auto&& draw) {
      const int a = draw(0, first ? (uint32_t)(W - 2) : 3u);
      const int exit_wall = a >= wall ? a + 1 : a, next_entry = first ? 2 : (exit_wall + 2) & 3;
      const bool ew_x = (exit_wall & 1) == 0;
      const int b = draw(1, first ? (uint32_t)(W - 2) : (uint32_t)((ew_x ? psy : psx) - 2));
      const int d = 1 + b;
      const int ex = first ? a : exit_wall == 0 ? ptx + psx - 1 : exit_wall == 2 ? ptx : ptx + d;
      const int ey = first ? b : exit_wall == 1 ? pty + psy - 1 : exit_wall == 3 ? pty : pty + d;
      int sx = 4, sy = 4;
      if (nsz) { sx += draw(2, rsz); sy += draw(3, rsz); }
      int tx = ex, ty = ey;
      if (!first) {
        const bool ne_x = (next_entry & 1) == 0;
        const int t = draw(2 + nsz, (uint32_t)((ne_x ? sy : sx) - 2));
        tx = next_entry == 0 ? ex - sx + 1 : next_entry == 2 ? ex : ex - sx + 2 + t;
        ty = next_entry == 1 ? ey - sy + 1 : next_entry == 3 ? ey : ey - sy + 2 + t;
      }
      bool fits = tx >= 0 && ty >= 0 && tx + sx <= W && ty + sy < H;
      for (int q = 0; q + 1 < n; q++) {                    // roomList[:-1]
        const uint32_t r = cur[q];
        const int rtx = (int)(r & 31u), rty = (int)((r >> 5) & 31u), rsx = (int)((r >> 10) & 15u), rsy = (int)((r >> 14) & 15u);
        fits = fits && (tx + sx < rtx || rtx + rsx <= tx || ty + sy < rty || rty + rsy <= ty);
      }
      ok = fits; room = mr_pack(tx, ty, sx, sy, ex, ey); entry = next_entry;
    };
    const R saved = rng;
    uint32_t w[5];
    next32_block(rng, (first ? 2 : 3) + nsz, w);
    bool unsafe = false;
    eval([&](int k, uint32_t r) {
      const uint32_t word = k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : k == 3 ? w[3] : w[4];
      const uint64_t m = (uint64_t)word * r;
      unsafe = unsafe || ((uint32_t)m < r && (r & (r - 1u)) != 0u);
#ifdef MG_MR_TEST_UNSAFE      // (test builds: most tries take the rand_int path from the saved stream position -- tests/test_generators_cpu.py)
      unsafe = unsafe || (word & 3u) != 0u;
#endif
      return (int)(m >> 32);
    });
    if (__builtin_expect(unsafe, 0)) {
      rng = saved;
      eval([&](int, uint32_t r) { return rand_int(rng, 0, (int)r); });
    }
    if (ok) { cur[n] = room; n++; wall = entry; i = 0; parent = room; }
    else if (!first) i++;
    const bool end = n >= num_rooms || (first ? !ok : i >= 8);
    if (end) {
      if (n > nbest) {
#pragma unroll
        for (int k = 0; k < 6; k++) bw[k] = k < n ? cur[k] : bw[k];
        nbest = n;
      }
      n = 0; wall = 2; i = 0;
    }
  }
}
template <class R, class G>
MG_HD void gen_multiroom(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int W = g.W;
  uint32_t* cur = (uint32_t*)g.p;               // the chain under construction (<= 6 words)
  MrState<G::kWave> state(g.p, P.scratch_off);
  uint32_t* st = state.p;                       // [0] numRooms, [1] rooms in the best chain, [2..7] the best chain
  int num_rooms, nbest;
  uint32_t bw[6] = { 0u, 0u, 0u, 0u, 0u, 0u };  // the lane form's best chain (registers: the grid it is built in is cleared below)
  if (!out.resume) {
    num_rooms = rand_int(rng, P.num_crossings, P.num_dists + 1);
    nbest = 0;
    st[0] = (uint32_t)num_rooms; st[1] = 0u;
  } else { num_rooms = (int)mr_word<G>(st[0]); nbest = (int)mr_word<G>(st[1]); }
  if constexpr (!G::kWave) mr_search_lane(rng, g, P, num_rooms, bw, nbest);
  auto best_room = [&](int idx) -> uint32_t {
    if constexpr (G::kWave) return mr_word<G>(st[2 + idx]);
    else return idx == 0 ? bw[0] : idx == 1 ? bw[1] : idx == 2 ? bw[2] : idx == 3 ? bw[3] : idx == 4 ? bw[4] : bw[5];
  };
  if constexpr (G::kWave)
  while (nbest < num_rooms && !rng.dead()) {
    MG_GA_ATTEMPT(out);
    rng.checkpoint();                           // st[] is consistent with the stream position here
    int n = 0, wall = 2;                        // entryDoorWall of the newest room
    const int ex = rand_int(rng, 0, W - 2), ey = rand_int(rng, 0, W - 2);
    if (mr_try_room(rng, g, P, cur, n, wall, ex, ey)) {
      while (n < num_rooms && !rng.dead()) {
        const uint32_t r = mr_word<G>(cur[n - 1]);
        const int tx = (int)(r & 31u), ty = (int)((r >> 5) & 31u), sx = (int)((r >> 10) & 15u), sy = (int)((r >> 14) & 15u);
        bool placed = false;
#pragma unroll 1
        for (int i = 0; i < 8 && !placed && !rng.dead(); i++) {
          if constexpr (G::kWave) {
            MG_WAVE_LDS_SYNC();                                     // (cur[] was written by this wave: the lanes read it below)
            uint32_t room = 0; int entry = 0;
            const int cnt = mr_spec(rng, g, P, cur, n, wall, 8 - i, placed, room, entry);
            if (cnt > 0) {
              i += cnt - 1;
              if (placed) { cur[n++] = room; wall = entry; }
              continue;
            }
          }
          const int k = rand_int(rng, 0, 3);                       // _rand_elem(sorted({0,1,2,3} - {entryDoorWall}))
          const int exit_wall = k >= wall ? k + 1 : k, next_entry = (exit_wall + 2) & 3;
          int dx, dy;
          if (exit_wall == 0) { dx = tx + sx - 1; dy = ty + rand_int(rng, 1, sy - 1); }
          else if (exit_wall == 1) { dx = tx + rand_int(rng, 1, sx - 1); dy = ty + sy - 1; }
          else if (exit_wall == 2) { dx = tx; dy = ty + rand_int(rng, 1, sy - 1); }
          else { dx = tx + rand_int(rng, 1, sx - 1); dy = ty; }
          if (mr_try_room(rng, g, P, cur, n, next_entry, dx, dy)) { placed = true; wall = next_entry; }
        }
        if (!placed) break;
      }
    }
    if (rng.dead()) { MG_GA(out, 8); return; }  // out of buffered draws inside this attempt: restart it from the checkpoint
    if (n > nbest) {
#pragma unroll 1
      for (int k = 0; k < n; k++) st[2 + k] = mr_word<G>(cur[k]);
      nbest = n; st[1] = (uint32_t)n;
    }
  }
  MG_GA(out, 8);
  if (rng.dead()) return;
  rng.checkpoint();                             // the chain is final: from here on only the drawing below is replayed
  g.clear_empty();
  uint32_t prev = 6u;                           // COLOR_NAMES index of the previous door, 6 = none yet
#pragma unroll 1
  for (int idx = 0; idx < nbest; idx++) {
    const uint32_t r = best_room(idx);
    const int tx = (int)(r & 31u), ty = (int)((r >> 5) & 31u), sx = (int)((r >> 10) & 15u), sy = (int)((r >> 14) & 15u);
    if constexpr (G::kWave) {
      // (the room's four walls, lane = position along the wall; in room order like the reference: a later room's walls go over an earlier door)
      MG_WAVE_LDS_SYNC();
      if (g.lane < sx) { g.p[ty * W + tx + g.lane] = (uint8_t)CELL_WALL_GREY; g.p[(ty + sy - 1) * W + tx + g.lane] = (uint8_t)CELL_WALL_GREY; }
      if (g.lane < sy) { g.p[(ty + g.lane) * W + tx] = (uint8_t)CELL_WALL_GREY; g.p[(ty + g.lane) * W + tx + sx - 1] = (uint8_t)CELL_WALL_GREY; }
      MG_WAVE_LDS_SYNC();
    } else {
    for (int i = 0; i < sx; i++) { g.set(tx + i, ty, CELL_WALL_GREY); g.set(tx + i, ty + sy - 1, CELL_WALL_GREY); }
    for (int j = 0; j < sy; j++) { g.set(tx, ty + j, CELL_WALL_GREY); g.set(tx + sx - 1, ty + j, CELL_WALL_GREY); }
    }
    if (idx > 0) {
      const uint32_t k = (uint32_t)rand_int(rng, 0, prev < 6u ? 5 : 6);        // sorted(COLOR_NAMES - {prevDoorColor})
      const uint32_t c = (prev < 6u && k >= prev) ? k + 1u : k;
      g.set((int)((r >> 18) & 31u), (int)((r >> 23) & 31u), make_cell(T_DOOR_CLOSED, color_from_sorted(c)));
      prev = c;
    }
  }
  MG_GA(out, 9);
  const uint32_t r0 = best_room(0), rl = best_room(nbest - 1);
  if (!place_agent(rng, g, (int)(r0 & 31u), (int)((r0 >> 5) & 31u), (int)((r0 >> 10) & 15u), (int)((r0 >> 14) & 15u), -1, out)) out.failed = true;
  int x, y;
  if (!place_obj(rng, g, CELL_GOAL, (int)(rl & 31u), (int)((rl >> 5) & 31u), (int)((rl >> 10) & 15u), (int)((rl >> 14) & 15u),
                 (int)out.ax, (int)out.ay, false, -1, x, y)) out.failed = true;
  out.mission = 0;
  MG_GA(out, 2);
}


// ---- the multi-room BabyAI levels with one instruction about one described object (first multi-room slice) ----
// envs/babyai/goto.py:403-426 (GoTo, GoToOpen, GoToObjMaze*; P.num_crossings = doors_open), pickup.py:66-72 (Pickup), open.py:69-86
// (Open) on RoomGrid._gen_grid / place_agent(random room) / connect_all / add_distractors(all_unique=False, random rooms) /
// check_objs_reachable, inside RoomGridLevel's retry loop (roomgrid_level.py:119-144).  nc x nr rooms (2 x 2 or 3 x 3) of size
// P.room_size on a grid of up to 22 x 22.  Mission ids: GoTo as GoToObj (article * 18 + colour * 3 + type); Pickup in the
// "pick up" table (article * 28 + (colour + 1) * 4 + type + 1); Open = article * 6 + colour.
//
// check_objs_reachable (roomgrid_level.py:250-302) on a grid wider than one 64-bit board: lane y holds ROW y as bit masks
// (passable = None or any door; objects = every cell that is neither None nor a wall, doors included; reached).  One flood iteration = step one row up / down
// (two lane shuffles) + an occluded fill along the row (Kogge-Stone, like vis_row), repeated until no row changes; an object
// is reachable when it lies in or next to the reached set.
template <class G>
MG_HD bool maze_objs_reachable(G& g, int ax, int ay) {
  if constexpr (!G::kWave) {
    // (one lane: the same row bitboards, all H rows in this lane; pinned on the CPU, not yet used on the device)
    const int W = g.W, H = g.H;
    uint32_t pass[25], obj[25], R[25];
    for (int y = 0; y < H; y++) {
      uint32_t pm = 0, om = 0;
      for (int x = 0; x < W; x++) {
        const uint32_t c = g.p[y * W + x], t = cell_type(c);
        const bool p = c == CELL_EMPTY || t == T_DOOR || t == T_DOOR_CLOSED || t == T_DOOR_LOCKED;
        pm |= (uint32_t)p << x;
        om |= (uint32_t)(c != CELL_EMPTY && t != T_WALL) << x;
      }
      pass[y] = pm; obj[y] = om; R[y] = y == ay ? 1u << ax : 0u;
    }
    for (int it = 0; it < 1024; it++) {
      bool changed = false;
      for (int y = 0; y < H; y++) {
        const uint32_t up = y > 0 ? R[y - 1] : 0u, dn = y + 1 < H ? R[y + 1] : 0u;
        const uint32_t seed = (R[y] | up | dn) & pass[y];
        uint32_t fr = seed, pr = pass[y], fl = seed, pl = pass[y];
        for (int s = 1; s < 32; s <<= 1) { fr |= pr & (fr << s); pr &= pr << s; fl |= pl & (fl >> s); pl &= pl >> s; }
        const uint32_t Rn = R[y] | fr | fl;
        changed |= Rn != R[y];
        R[y] = Rn;
      }
      if (!changed) break;
    }
    bool all_near = true;
    for (int y = 0; y < H; y++) {
      const uint32_t up = y > 0 ? R[y - 1] : 0u, dn = y + 1 < H ? R[y + 1] : 0u;
      const uint32_t near = R[y] | (R[y] << 1) | (R[y] >> 1) | up | dn;
      all_near = all_near && (obj[y] & ~near) == 0u;
    }
    return all_near;
  } else {
  MG_WAVE_LDS_SYNC();
  const int W = g.W, H = g.H;
  uint32_t pass = 0, obj = 0;
  if (g.lane < H)
    for (int x = 0; x < W; x++) {
      const uint32_t c = g.p[g.lane * W + x], t = cell_type(c);
      const bool p = c == CELL_EMPTY || t == T_DOOR || t == T_DOOR_CLOSED || t == T_DOOR_LOCKED;
      pass |= (uint32_t)p << x;
      obj |= (uint32_t)(c != CELL_EMPTY && t != T_WALL) << x;      // doors have to be reached as well (roomgrid_level.py:292-300)
    }
  uint32_t R = g.lane == ay ? 1u << ax : 0u;
  for (int it = 0; it < 1024; it++) {
    uint32_t up = (uint32_t)__shfl_up((int)R, 1), dn = (uint32_t)__shfl_down((int)R, 1);
    if (g.lane == 0) up = 0u;
    if (g.lane == 63) dn = 0u;
    const uint32_t seed = (R | up | dn) & pass;
    uint32_t fr = seed, pr = pass, fl = seed, pl = pass;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) { fr |= pr & (fr << s); pr &= pr << s; fl |= pl & (fl >> s); pl &= pl >> s; }
    const uint32_t Rn = R | fr | fl;
    const bool changed = __ballot(Rn != R) != 0ull;
    R = Rn;
    if (!changed) break;
  }
  uint32_t up = (uint32_t)__shfl_up((int)R, 1), dn = (uint32_t)__shfl_down((int)R, 1);
  if (g.lane == 0) up = 0u;
  if (g.lane == 63) dn = 0u;
  const uint32_t near = R | (R << 1) | (R >> 1) | up | dn;
  return __ballot((obj & ~near) != 0u) == 0ull;
  }
}
enum : int { KIND_BABYAI_GOTO = 33, KIND_BABYAI_PICKUP = 34, KIND_BABYAI_OPEN = 35 };
MG_HD uint32_t sorted_from_color(uint32_t c) {       // inverse of color_from_sorted: COLOR_TO_IDX -> position in the sorted COLOR_NAMES
  const uint32_t packed = (4u << (4 * C_RED)) | (1u << (4 * C_GREEN)) | (0u << (4 * C_BLUE)) | (3u << (4 * C_PURPLE)) | (5u << (4 * C_YELLOW)) | (2u << (4 * C_GREY));
  return (packed >> (4u * c)) & 15u;
}
template <class R, class G>
MG_HD void gen_babyai_maze(R& rng, G& g, const GenParams& P, GenResult& out) {
  const int rs = P.room_size, W = g.W, H = g.H, st = rs - 1;
  const int nc = (W - 1) / st, nr = (H - 1) / st, nrooms = nc * nr;
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    MG_GA_ATTEMPT(out);
    rng.checkpoint();                           // RecursionError / RejectSampling regenerate from the current stream position
    if constexpr (G::kWave) {
    g.lattice(st);
    } else {
      g.clear_empty();
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if ((x % st) == 0 || (y % st) == 0) g.set(x, y, CELL_WALL_GREY);
    }
    uint64_t right_y = 0, down_x = 0, doors = 0;     // nibble per room (r = j * nc + i): door offsets inside the room; bit 4r+k: room r has a door on side k
#pragma unroll 1
    for (int j = 0; j < nr; j++)
#pragma unroll 1
      for (int i = 0; i < nc; i++) {
        const int r = j * nc + i, tx = i * st, ty = j * st;
        if (i < nc - 1) right_y |= (uint64_t)(rand_int(rng, ty + 1, ty + rs - 1) - ty) << (4 * r);
        if (j < nr - 1) down_x |= (uint64_t)(rand_int(rng, tx + 1, tx + rs - 1) - tx) << (4 * r);
      }
    // RoomGrid.place_agent(i=None, j=None) (roomgrid.py:313-334): a random room, then until the front cell is free
    MG_GA(out, 1);
    const int ai = rand_int(rng, 0, nc), aj = rand_int(rng, 0, nr);
    const bool agent_ok = rg_place_agent(rng, g, ai * st, aj * st, rs, out);
    MG_GA(out, 2);
    if (!agent_ok) continue;
    const int ax = (int)out.ax, ay = (int)out.ay;
    // connect_all (roomgrid.py:336-394)
    const int start = (ay / st) * nc + ax / st;
    bool fail = false;
    auto door_xy = [&](int i, int j, int k, int& dx, int& dy) {      // Room.door_pos[k]: the left / up door is the neighbour's right / down door
      const int ri = k == 2 ? i - 1 : i, rj = k == 3 ? j - 1 : j, rr = rj * nc + ri;
      const bool vertical_wall = k == 0 || k == 2;
      dx = vertical_wall ? ri * st + st : ri * st + (int)((down_x >> (4 * rr)) & 15u);
      dy = vertical_wall ? rj * st + (int)((right_y >> (4 * rr)) & 15u) : rj * st + st;
    };
    bool connected = nrooms == 1;                  // (no door yet: only a one-room grid is connected)
    for (int itr = 0; !rng.dead(); itr++) {
      if (itr > 5000) { fail = true; break; }
      if (connected) break;
      int i = 0, j = 0, k = 0;
      bool picked = false;
      if constexpr (G::kWave) {
        const int n = connect_spec(rng, g, nc, nr, doors, 0u, 5001 - itr, picked, i, j, k);
        if (n > 0) { itr += n - 1; if (!picked) continue; }
      }
      if (!picked) {
        i = rand_int(rng, 0, nc); j = rand_int(rng, 0, nr); k = rand_int(rng, 0, 4);
        const bool has_nb = k == 0 ? i < nc - 1 : k == 1 ? j < nr - 1 : k == 2 ? i > 0 : j > 0;
        if (!has_nb || ((doors >> ((j * nc + i) * 4 + k)) & 1ull)) continue;
      }
      const int r = j * nc + i;
      const uint32_t dc = (uint32_t)rand_int(rng, 0, 6);
      int dx, dy;
      door_xy(i, j, k, dx, dy);
      g.set(dx, dy, make_cell(T_DOOR_CLOSED, color_from_sorted(dc)));
      const int nrm = r + (k == 0 ? 1 : k == 1 ? nc : k == 2 ? -1 : -nc);
      doors |= (1ull << (r * 4 + k)) | (1ull << (nrm * 4 + ((k + 2) & 3)));
      connected = rooms_reach(doors, start, nc, nrooms) == rooms_all(nrooms);      // find_reach, after the one thing that changes it
    }
    MG_GA(out, 3);
    if (fail || rng.dead()) continue;
    // add_distractors(num_distractors, all_unique=False) (roomgrid.py:396-438): colour, type, then a random room; reject_next_to
    // sees the agent where it now stands
    uint64_t ocol = 0, otyp = 0;                 // 3 / 2 bits per object
    const int nd = min(P.num_dists, 21);
    bool ok = true;
    int x, y;
#pragma unroll 1
    for (int n = 0; n < nd && ok && !rng.dead(); n++) {
      uint32_t ci, ti;
      int ri, rj, v4[4];
      bool four = false;
      if constexpr (G::kWave) four = draw4_spec(rng, g, 6u, 3u, (uint32_t)nc, (uint32_t)nr, v4);      // colour, type, room column, room row: one pass
      if (four) { ci = (uint32_t)v4[0]; ti = (uint32_t)v4[1]; ri = v4[2]; rj = v4[3]; }
      else { ci = (uint32_t)rand_int(rng, 0, 6); ti = (uint32_t)rand_int(rng, 0, 3); ri = rand_int(rng, 0, nc); rj = rand_int(rng, 0, nr); }
      ok = place_obj(rng, g, make_cell((uint32_t)T_KEY + ti, color_from_sorted(ci)), ri * st, rj * st, rs, rs, ax, ay, true, 1000, x, y);
      ocol |= (uint64_t)ci << (3 * n); otyp |= (uint64_t)ti << (2 * n);
    }
    MG_GA(out, 4);
    if (!ok || rng.dead()) continue;
    const bool reachable = maze_objs_reachable(g, ax, ay);
    MG_GA(out, 5);
    if (!reachable) continue;                                        // RejectSampling
    if (P.kind == KIND_BABYAI_OPEN) {
      // Open.gen_mission (open.py:69-86): every room's doors in (column, row, right/down/left/up) order -- each door once per side --
      // one of them picked; the description is its colour; "a" when another door has that colour
      int cnt = 0;
      for (int r = 0; r < nrooms; r++) cnt += __builtin_popcount((uint32_t)(doors >> (4 * r)) & 15u);
      const int pick = rand_int(rng, 0, cnt);
      uint32_t pc = 0;
      int seen = 0;
#pragma unroll 1
      for (int i = 0; i < nc; i++)
#pragma unroll 1
        for (int j = 0; j < nr; j++)
#pragma unroll 1
          for (int k = 0; k < 4; k++)
            if ((doors >> ((j * nc + i) * 4 + k)) & 1ull) {
              if (seen == pick) { int dx, dy; door_xy(i, j, k, dx, dy); pc = cell_color(g.get(dx, dy)); }
              seen++;
            }
      uint32_t same = 0;
      if constexpr (G::kWave) {
      MG_WAVE_LDS_SYNC();
      for (int base = 0; base < W * H; base += 64) {
        const int q = base + g.lane;
        const uint32_t c = q < W * H ? (uint32_t)g.p[q] : 0u;
        same += (uint32_t)__popcll(__ballot(q < W * H && cell_ref_type(c) == T_DOOR && cell_type(c) != T_BOX_KEY && cell_color(c) == pc));
      }
      } else {
        for (int q = 0; q < W * H; q++) { const uint32_t c = (uint32_t)g.p[q]; same += (cell_ref_type(c) == T_DOOR && cell_type(c) != T_BOX_KEY && cell_color(c) == pc) ? 1u : 0u; }
      }
      out.mission = (same > 1u ? 6u : 0u) + sorted_from_color(pc);
      out.aux = ~0ull;
      MG_GA(out, 6);
      return;
    }
    const int k = rand_int(rng, 0, nd);
    const uint32_t kc = (uint32_t)(ocol >> (3 * k)) & 7u, kt = (uint32_t)(otyp >> (2 * k)) & 3u;
    uint32_t matches = 0;
    for (int n = 0; n < nd; n++) matches += (((uint32_t)(ocol >> (3 * n)) & 7u) == kc && ((uint32_t)(otyp >> (2 * n)) & 3u) == kt) ? 1u : 0u;
    if (P.kind == KIND_BABYAI_GOTO) {
      out.mission = (matches > 1u ? 18u : 0u) + kc * 3u + kt;
      if (P.num_crossings) {                                         // open_all_doors (roomgrid_level.py:238-248)
        if constexpr (G::kWave) {
        MG_WAVE_LDS_SYNC();
        for (int base = 0; base < W * H; base += 64) {
          const int q = base + g.lane;
          if (q < W * H && cell_type(g.p[q]) == T_DOOR_CLOSED) g.p[q] = (uint8_t)make_cell(T_DOOR, cell_color(g.p[q]));
        }
        MG_WAVE_LDS_SYNC();
        } else {
          for (int q = 0; q < W * H; q++) if (cell_type(g.p[q]) == T_DOOR_CLOSED) g.p[q] = (uint8_t)make_cell(T_DOOR, cell_color(g.p[q]));
        }
      }
    } else {
      out.mission = (matches > 1u ? 28u : 0u) + (kc + 1u) * 4u + (kt + 1u);
    }
    out.aux = ~0ull;                            // GoToInstr on a large grid: no stale tracked position yet (see k_step, RULE_GOTO_BIG)
    MG_GA(out, 6);
    return;
  }
  out.failed = true;
}

// ---- general RoomGrid helpers for the BabyAI levels below (core/roomgrid.py; the oracle's rg_* functions are the same restated in C) ----
// nc x nr rooms of size rs; nibble r of right_off / down_off = offset of room r's right / down door position inside the room
// (Room.door_pos, roomgrid.py:158-171), bit 4r+k of doors = room r has a door (or no wall) on side k = right, down, left, up;
// bit r of locked = Room.locked.  (ax, ay) = env.agent_pos as reject_next_to sees it: the middle of the grid until place_agent.
struct RG {
  int rs, st, nc, nr, ax, ay;
  uint64_t right_off, down_off, doors;
  uint32_t locked;
  bool ok;                                          // false after a RecursionError (place_obj ran out of tries): regenerate
  MG_HD int room(int i, int j) const { return j * nc + i; }
  MG_HD bool has_nb(int i, int j, int k) const { return k == 0 ? i < nc - 1 : k == 1 ? j < nr - 1 : k == 2 ? i > 0 : j > 0; }
  MG_HD void door_xy(int i, int j, int k, int& dx, int& dy) const {
    const int ri = k == 2 ? i - 1 : i, rj = k == 3 ? j - 1 : j, rr = rj * nc + ri;
    const bool vertical_wall = k == 0 || k == 2;
    // MG_SHC: add_door may arrive here with a side that has NO neighbour when the stream's draw budget ran out (its loop leaves on rng.dead()): rr / nrm
    // are negative then, the shift count is negative, and the episode is discarded and redrawn.  The hardware masks a 64-bit shift count to six bits;
    // in C++ the shift is undefined -- found by UBSan on the emulator (profiles/r4/emu_all_ids_address_undefined.txt).  The count is masked explicitly,
    // on the device too since round 5 (the masked form ran the RoomGrid / BabyAI GPU tests as a variant build first: 1 415 passed, same refill rates --
    // profiles/r5/shift_mask_*.txt).
#define MG_SHC(x) ((x) & 63)
    dx = vertical_wall ? ri * st + st : ri * st + (int)((down_off >> MG_SHC(4 * rr)) & 15u);
    dy = vertical_wall ? rj * st + (int)((right_off >> MG_SHC(4 * rr)) & 15u) : rj * st + st;
  }
  MG_HD void mark(int i, int j, int k) {
    const int r = room(i, j), nrm = r + (k == 0 ? 1 : k == 1 ? nc : k == 2 ? -1 : -nc);
    doors |= (1ull << (r * 4 + k)) | (1ull << MG_SHC(nrm * 4 + ((k + 2) & 3)));
  }
#undef MG_SHC
  // RoomGrid._gen_grid (roomgrid.py:123-179)
  template <class R, class G> MG_HD void gen_grid(R& rng, G& g, int room_size) {
    rs = room_size; st = rs - 1; nc = (g.W - 1) / st; nr = (g.H - 1) / st;
    right_off = 0; down_off = 0; doors = 0; locked = 0; ok = true;
    if constexpr (G::kWave) {
    g.lattice(st);
    } else {
      g.clear_empty();
      for (int y = 0; y < g.H; y++) for (int x = 0; x < g.W; x++) if ((x % st) == 0 || (y % st) == 0) g.set(x, y, CELL_WALL_GREY);
    }
#pragma unroll 1
    for (int j = 0; j < nr; j++)
#pragma unroll 1
      for (int i = 0; i < nc; i++) {
        const int r = room(i, j), tx = i * st, ty = j * st;
        if (i < nc - 1) right_off |= (uint64_t)(rand_int(rng, ty + 1, ty + rs - 1) - ty) << (4 * r);
        if (j < nr - 1) down_off |= (uint64_t)(rand_int(rng, tx + 1, tx + rs - 1) - tx) << (4 * r);
      }
    ax = (nc / 2) * st + rs / 2; ay = (nr / 2) * st + rs / 2;
  }
  // RoomGrid.add_door (roomgrid.py:230-277); k / ci / locked < 0 = draw it.  Returns the COLOR_NAMES index used.
  template <class R, class G> MG_HD int add_door(R& rng, G& g, int i, int j, int k, int ci, int lock, int& dx, int& dy) {
    if (k < 0)
      for (;;) { k = rand_int(rng, 0, 4); if (rng.dead() || (has_nb(i, j, k) && !((doors >> (room(i, j) * 4 + k)) & 1ull))) break; }
    if (ci < 0) ci = rand_int(rng, 0, 6);                         // _rand_color()
    if (lock < 0) lock = rand_int(rng, 0, 2) == 0;                // _rand_bool()
    if (lock) locked |= 1u << room(i, j);
    else locked &= ~(1u << room(i, j));                           // (add_door assigns room.locked = locked)
    door_xy(i, j, k, dx, dy);
    g.set(dx, dy, make_cell(lock ? (uint32_t)T_DOOR_LOCKED : (uint32_t)T_DOOR_CLOSED, color_from_sorted((uint32_t)ci)));
    mark(i, j, k);
    return ci;
  }
  // RoomGrid.add_object / place_in_room (roomgrid.py:181-228); ti / ci < 0 = draw (kind first, then colour)
  template <class R, class G> MG_HD void add_object(R& rng, G& g, int i, int j, int ti, int ci, int& ti_out, int& ci_out) {
    if (ti < 0) ti = rand_int(rng, 0, 3);
    if (ci < 0) ci = rand_int(rng, 0, 6);
    int x, y;
    if (!place_obj(rng, g, make_cell((uint32_t)T_KEY + (uint32_t)ti, color_from_sorted((uint32_t)ci)), i * st, j * st, rs, rs, ax, ay, true, 1000, x, y)) ok = false;
    ti_out = ti; ci_out = ci;
  }
  // RoomGrid.place_agent(i, j) (roomgrid.py:313-334)
  template <class R, class G> MG_HD void place_agent_in(R& rng, G& g, int i, int j, GenResult& out) {
    if (!rg_place_agent(rng, g, i * st, j * st, rs, out)) { ok = false; return; }
    ax = (int)out.ax; ay = (int)out.ay;
  }
  // RoomGrid.connect_all (roomgrid.py:336-394) with door_colors = COLOR_NAMES without index `exclude` (< 0: all six)
  template <class R, class G> MG_HD void connect_all(R& rng, G& g, int exclude) {
    const int start = (ay / st) * nc + ax / st, nrooms = nc * nr;
    // (find_reach is a function of the doors alone: recomputed after a door was added, not in every iteration -- see rooms_reach)
    bool connected = rooms_reach(doors, start, nc, nrooms) == rooms_all(nrooms);
    for (int itr = 0; !rng.dead(); itr++) {
      if (itr > 5000) { ok = false; return; }
      if (connected) return;
      int i = 0, j = 0, k = 0;
      bool picked = false;
      if constexpr (G::kWave) {
        const int n = connect_spec(rng, g, nc, nr, doors, locked, 5001 - itr, picked, i, j, k);
        if (n > 0) { itr += n - 1; if (!picked) continue; }
      }
      if (!picked) {
        i = rand_int(rng, 0, nc); j = rand_int(rng, 0, nr); k = rand_int(rng, 0, 4);
        if (!has_nb(i, j, k) || ((doors >> (room(i, j) * 4 + k)) & 1ull)) continue;
        const int ni = i + (k == 0) - (k == 2), nj = j + (k == 1) - (k == 3);
        if (((locked >> room(i, j)) | (locked >> room(ni, nj))) & 1u) continue;
      }
      int ci = rand_int(rng, 0, exclude >= 0 ? 5 : 6), dx, dy;
      if (exclude >= 0 && ci >= exclude) ci++;
      const uint32_t keep = locked;                                // connect_all's add_door(..., locked=False) on an unlocked room
      add_door(rng, g, i, j, k, ci, 0, dx, dy);
      locked = keep;
      connected = rooms_reach(doors, start, nc, nrooms) == rooms_all(nrooms);
    }
  }
};
// number of cells on the grid that hold a door of COLOR_TO_IDX colour c / exactly the cell code `code`
template <class G>
MG_HD uint32_t count_cells(G& g, bool doors_of_color, uint32_t c) {
  if constexpr (!G::kWave) {
    uint32_t n = 0;
    for (int q = 0; q < g.W * g.H; q++) {
      const uint32_t v = (uint32_t)g.p[q];
      n += (doors_of_color ? (cell_ref_type(v) == T_DOOR && cell_color(v) == c) : v == c) ? 1u : 0u;
    }
    return n;
  } else {
  MG_WAVE_LDS_SYNC();
  uint32_t n = 0;
  for (int base = 0; base < g.W * g.H; base += 64) {
    const int q = base + g.lane;
    const uint32_t v = q < g.W * g.H ? (uint32_t)g.p[q] : 0u;
    const bool hit = q < g.W * g.H && (doors_of_color ? (cell_ref_type(v) == T_DOOR && cell_color(v) == c) : v == c);
    n += (uint32_t)__popcll(__ballot(hit));
  }
  return n;
  }
}

enum : int { KIND_BABYAI_UNLOCKPICKUP = 36, KIND_BABYAI_BLOCKEDUNLOCKPICKUP = 37, KIND_UNLOCKTOUNLOCK = 38, KIND_KEYINBOX = 39, KIND_BABYAI_UNLOCK = 40,
             KIND_BABYAI_GOTODOOR = 41, KIND_GOTOOBJDOOR = 42, KIND_UNBLOCKPICKUP = 43, KIND_PICKUPABOVE = 44, KIND_GOTOIMPUNLOCK = 45 };
// envs/babyai/unlock.py: UnlockPickup (:307-319; P.num_dists = 0 | 4 distractors = UnlockPickupDist), BlockedUnlockPickup (:380-393),
// UnlockToUnlock (:452-474), Unlock (:67-112); goto.py: GoToDoor (:730-740), GoToObjDoor (:800-813), GoToImpUnlock (:486-531);
// pickup.py: UnblockPickup (:128-140), PickupAbove (:354-362).  One PickupInstr / OpenInstr / GoToInstr about one description.
// Mission ids: "pick up" table (article * 28 + (colour + 1) * 4 + type + 1); Unlock / GoToDoor article * 6 + colour;
// GoToObjDoor article * 24 + colour * 4 + (key, ball, box, door); GoToImpUnlock as GoToObj.
template <class R, class G>
MG_HD void gen_babyai_levels(R& rng, G& g, const GenParams& P, GenResult& out) {
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();
    RG rg;
    rg.gen_grid(rng, g, P.room_size);
    int dx, dy, ti, ci;
    out.aux = ~0ull;
    if (P.kind == KIND_BABYAI_UNLOCKPICKUP) {
      int oc, dc;
      rg.add_object(rng, g, 1, 0, 2, -1, ti, oc);                             // the box
      dc = rg.add_door(rng, g, 0, 0, 0, -1, 1, dx, dy);
      rg.add_object(rng, g, 0, 0, 0, dc, ti, ci);                             // its key
      uint32_t used = (1u << (oc * 3 + 2)) | (1u << (dc * 3 + 0));           // add_distractors(num_distractors, all_unique): random rooms
      for (int n = 0; n < P.num_dists && rg.ok && !rng.dead();) {
        const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
        if ((used >> (c2 * 3 + t2)) & 1u) continue;
        const int ri = rand_int(rng, 0, rg.nc), rj = rand_int(rng, 0, rg.nr);
        rg.add_object(rng, g, ri, rj, t2, c2, ti, ci);
        used |= 1u << (c2 * 3 + t2); n++;
      }
      if (!rg.ok || rng.dead()) continue;
      rg.place_agent_in(rng, g, 0, 0, out);
      if (!rg.ok || rng.dead()) continue;
      out.mission = (uint32_t)(oc + 1) * 4u + 3u;                             // "pick up the {colour} box"
      return;
    }
    if (P.kind == KIND_BABYAI_BLOCKEDUNLOCKPICKUP) {
      int oc;
      rg.add_object(rng, g, 1, 0, 2, -1, ti, oc);
      const int dc = rg.add_door(rng, g, 0, 0, 0, -1, 1, dx, dy);
      const int bc = rand_int(rng, 0, 6);
      g.set(dx - 1, dy, make_cell(T_BALL, color_from_sorted((uint32_t)bc)));
      rg.add_object(rng, g, 0, 0, 0, dc, ti, ci);
      rg.place_agent_in(rng, g, 0, 0, out);
      if (!rg.ok || rng.dead()) continue;
      out.mission = 3u;                                                       // "pick up the box"
      return;
    }
    if (P.kind == KIND_UNLOCKTOUNLOCK) {
      uint32_t avail = 0x543210u;                                             // _rand_subset(COLOR_NAMES, 2)
      int colors[2];
      for (int n = 0, na = 6; n < 2; n++, na--) {
        const int k = rand_int(rng, 0, na);
        colors[n] = (int)((avail >> (4 * k)) & 15u);
        const uint32_t lowmask = (1u << (4 * k)) - 1u;
        avail = (avail & lowmask) | ((avail >> 4) & ~lowmask);
      }
      rg.add_door(rng, g, 0, 0, 0, colors[0], 1, dx, dy);
      rg.add_object(rng, g, 2, 0, 0, colors[0], ti, ci);
      rg.add_door(rng, g, 1, 0, 0, colors[1], 1, dx, dy);
      rg.add_object(rng, g, 1, 0, 0, colors[1], ti, ci);
      rg.add_object(rng, g, 0, 0, 1, -1, ti, ci);
      rg.place_agent_in(rng, g, 1, 0, out);
      if (!rg.ok || rng.dead()) continue;
      out.mission = 2u;                                                       // "pick up the ball"
      return;
    }
    if (P.kind == KIND_KEYINBOX) {
      // KeyInBox (unlock.py:232-242): a locked door, its key inside a box of a random colour (Box(colour, contains=Key(door colour)):
      // cell type T_BOX_DOORKEY, the key's colour is read off the door when the box is opened)
      rg.add_door(rng, g, 1, 1, -1, -1, 1, dx, dy);
      const int bc = rand_int(rng, 0, 6);
      int x, y;
      if (!place_obj(rng, g, make_cell(T_BOX_DOORKEY, color_from_sorted((uint32_t)bc)), rg.st, rg.st, rg.rs, rg.rs, rg.ax, rg.ay, true, 1000, x, y)) continue;
      rg.place_agent_in(rng, g, 1, 1, out);
      if (!rg.ok || rng.dead()) continue;
      out.mission = 0;
      return;
    }
    if (P.kind == KIND_BABYAI_GOTODOOR || P.kind == KIND_GOTOOBJDOOR) {
      uint64_t ocol = 0, otyp = 0;                                            // 3 / 2 bits per entry; type index 3 = door
      int n = 0;
      if (P.kind == KIND_GOTOOBJDOOR) {
        rg.place_agent_in(rng, g, 1, 1, out);
        for (; n < 8 && rg.ok && !rng.dead(); n++) {                          // add_distractors(1, 1, 8, all_unique=False)
          const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
          rg.add_object(rng, g, 1, 1, t2, c2, ti, ci);
          ocol |= (uint64_t)c2 << (3 * n); otyp |= (uint64_t)t2 << (2 * n);
        }
        if (!rg.ok || rng.dead()) continue;
      }
      for (int d = 0; d < 4 && !rng.dead(); d++) {
        const int c2 = rg.add_door(rng, g, 1, 1, -1, -1, -1, dx, dy);
        ocol |= (uint64_t)c2 << (3 * n); otyp |= 3ull << (2 * n); n++;
      }
      if (P.kind == KIND_BABYAI_GOTODOOR) rg.place_agent_in(rng, g, 1, 1, out);
      if (!rg.ok || rng.dead()) continue;
      if (P.kind == KIND_GOTOOBJDOOR && !maze_objs_reachable(g, rg.ax, rg.ay)) continue;
      const int k = rand_int(rng, 0, n);
      const uint32_t kc = (uint32_t)(ocol >> (3 * k)) & 7u, kt = (uint32_t)(otyp >> (2 * k)) & 3u;
      const uint32_t nposs = kt == 3u ? count_cells(g, true, color_from_sorted(kc)) : count_cells(g, false, make_cell((uint32_t)T_KEY + kt, color_from_sorted(kc)));
      out.mission = P.kind == KIND_BABYAI_GOTODOOR ? (nposs > 1u ? 6u : 0u) + kc : (nposs > 1u ? 24u : 0u) + kc * 4u + kt;
      return;
    }
    if (P.kind == KIND_UNBLOCKPICKUP) {
      const int ai = rand_int(rng, 0, rg.nc), aj = rand_int(rng, 0, rg.nr);
      rg.place_agent_in(rng, g, ai, aj, out);
      if (!rg.ok || rng.dead()) continue;
      rg.connect_all(rng, g, -1);
      if (!rg.ok || rng.dead()) continue;
      uint64_t ocol = 0, otyp = 0;
      for (int n = 0; n < 20 && rg.ok && !rng.dead(); n++) {                  // add_distractors(num_distractors=20, all_unique=False)
        const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
        const int ri = rand_int(rng, 0, rg.nc), rj = rand_int(rng, 0, rg.nr);
        rg.add_object(rng, g, ri, rj, t2, c2, ti, ci);
        ocol |= (uint64_t)c2 << (3 * n); otyp |= (uint64_t)t2 << (2 * n);
      }
      if (!rg.ok || rng.dead()) continue;
      if (maze_objs_reachable(g, rg.ax, rg.ay)) continue;                     // RejectSampling("all objects reachable")
      const int k = rand_int(rng, 0, 20);
      const uint32_t kc = (uint32_t)(ocol >> (3 * k)) & 7u, kt = (uint32_t)(otyp >> (2 * k)) & 3u;
      uint32_t matches = 0;
      for (int d = 0; d < 20; d++) matches += (((uint32_t)(ocol >> (3 * d)) & 7u) == kc && ((uint32_t)(otyp >> (2 * d)) & 3u) == kt) ? 1u : 0u;
      out.mission = (matches > 1u ? 28u : 0u) + (kc + 1u) * 4u + (kt + 1u);
      return;
    }
    if (P.kind == KIND_PICKUPABOVE) {
      int ot, oc;
      rg.add_object(rng, g, 1, 0, -1, -1, ot, oc);
      rg.add_door(rng, g, 1, 1, 3, -1, 0, dx, dy);
      rg.place_agent_in(rng, g, 1, 1, out);
      if (!rg.ok || rng.dead()) continue;
      rg.connect_all(rng, g, -1);
      if (!rg.ok || rng.dead()) continue;
      out.mission = (uint32_t)(oc + 1) * 4u + (uint32_t)(ot + 1);
      return;
    }
    // Unlock (unlock.py:67-112) and GoToImpUnlock (goto.py:486-531): a locked room, its key anywhere (the reference compares numpy
    // integers by identity -- `ik is id` --, which is never true: the key may land in the locked room itself, and distractors go
    // into every room), distractors in every room, the agent outside the locked room
    const int id = rand_int(rng, 0, rg.nc), jd = rand_int(rng, 0, rg.nr);
    const int dc = rg.add_door(rng, g, id, jd, -1, -1, 1, dx, dy);
    { const int ik = rand_int(rng, 0, rg.nc), jk = rand_int(rng, 0, rg.nr); rg.add_object(rng, g, ik, jk, 0, dc, ti, ci); }
    if (!rg.ok || rng.dead()) continue;
    if (P.kind == KIND_BABYAI_UNLOCK) { if (rand_int(rng, 0, 2) == 0) rg.connect_all(rng, g, dc); else rg.connect_all(rng, g, -1); }      // _rand_bool()
    else rg.connect_all(rng, g, -1);
    if (!rg.ok || rng.dead()) continue;
    const int per_room = P.kind == KIND_BABYAI_UNLOCK ? 3 : 2;
#pragma unroll 1
    for (int i = 0; i < rg.nc && rg.ok && !rng.dead(); i++)
#pragma unroll 1
      for (int j = 0; j < rg.nr && rg.ok && !rng.dead(); j++)
        for (int n = 0; n < per_room && rg.ok && !rng.dead(); n++) {          // add_distractors(i, j, per_room, all_unique=False)
          const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
          rg.add_object(rng, g, i, j, t2, c2, ti, ci);
        }
    if (!rg.ok || rng.dead()) continue;
    for (;;) {                                                                // place_agent() until it is not in the locked room
      const int ai = rand_int(rng, 0, rg.nc), aj = rand_int(rng, 0, rg.nr);
      rg.place_agent_in(rng, g, ai, aj, out);
      if (!rg.ok || rng.dead()) break;
      if (rg.ax / rg.st == id && rg.ay / rg.st == jd) continue;
      break;
    }
    if (!rg.ok || rng.dead()) continue;
    if (!maze_objs_reachable(g, rg.ax, rg.ay)) continue;
    if (P.kind == KIND_BABYAI_UNLOCK) {
      const uint32_t count = count_cells(g, true, color_from_sorted((uint32_t)dc));
      out.mission = (count > 1u ? 6u : 0u) + (uint32_t)dc;
      return;
    }
    const int c3 = rand_int(rng, 0, 6), t3 = rand_int(rng, 0, 3);             // GoToImpUnlock: the target goes into the locked room
    rg.add_object(rng, g, id, jd, t3, c3, ti, ci);
    if (!rg.ok || rng.dead()) continue;
    const uint32_t nposs = count_cells(g, false, make_cell((uint32_t)T_KEY + (uint32_t)t3, color_from_sorted((uint32_t)c3)));
    out.mission = (nposs > 1u ? 18u : 0u) + (uint32_t)c3 * 3u + (uint32_t)t3;
    return;
  }
  out.failed = true;
}

enum : int { KIND_PUTNEXTLOCAL = 46, KIND_PUTNEXT = 47, KIND_ACTIONOBJDOOR = 48, KIND_OPENDOOR = 49 };
// envs/babyai/putnext.py: PutNextLocal (:72-80), PutNext (:168-214; P.num_crossings = start_carrying); other.py:86-106 ActionObjDoor;
// open.py:209-229 OpenDoor (P.num_crossings = select_by: 0 random, 1 colour, 2 location; the strict / debug variant differs in the
// step rule only).  Mission ids: PutNext (move colour * 3 + move type) * 18 + fixed colour * 3 + fixed type; ActionObjDoor
// verb * 48 + article * 24 + colour * 4 + (key, ball, box, door), verb = go to | pick up | open; OpenDoor colour (0..5) or
// 6 + article * 4 + (left, right, front, behind).  out.aux: PutNext = the cell the carried object came from (start_carrying);
// ActionObjDoor = no stale tracked position (RULE_GOTO_BIG); OpenDoor = COLOR_TO_IDX bit mask of the described doors.
template <class R, class G>
MG_HD void gen_babyai_put_open(R& rng, G& g, const GenParams& P, GenResult& out) {
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();
    RG rg;
    rg.gen_grid(rng, g, P.room_size);
    int dx, dy, ti, ci;
    out.aux = ~0ull; out.carry = 0;
    if (P.kind == KIND_PUTNEXTLOCAL || P.kind == KIND_PUTNEXT) {
      const int per = P.num_dists, rooms = P.kind == KIND_PUTNEXT ? 2 : 1;
      rg.place_agent_in(rng, g, 0, 0, out);                   // place_agent() of a 1 x 1 room grid draws no room index either
      uint64_t opos = 0;                                      // byte n = x | y << 4 of object n
      uint64_t okind = 0; uint32_t used = 0;                  // byte n = colour * 3 + type of object n; bit per used (colour, type)
      int n = 0;
      for (int room = 0; room < rooms && rg.ok && !rng.dead(); room++)
        for (int k = 0; k < per && rg.ok && !rng.dead();) {   // add_distractors(room, 0, per): unique over every room's objects
          const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
          if ((used >> (c2 * 3 + t2)) & 1u) continue;
          int x, y;
          if (!place_obj(rng, g, make_cell((uint32_t)T_KEY + (uint32_t)t2, color_from_sorted((uint32_t)c2)), room * rg.st, 0, rg.rs, rg.rs, rg.ax, rg.ay, true, 1000, x, y)) { rg.ok = false; break; }
          used |= 1u << (c2 * 3 + t2);
          if (n < 8) { opos |= (uint64_t)(x | (y << 4)) << (8 * n); okind |= (uint64_t)(c2 * 3 + t2) << (8 * n); }
          n++; k++;
        }
      if (!rg.ok || rng.dead()) continue;
      int a, b;
      if (P.kind == KIND_PUTNEXTLOCAL) {
        if (!maze_objs_reachable(g, rg.ax, rg.ay)) continue;
        a = rand_int(rng, 0, n);                              // _rand_subset(objs, 2)
        b = rand_int(rng, 0, n - 1); if (b >= a) b++;
      } else {
        if constexpr (G::kWave) {
        MG_WAVE_LDS_SYNC();                                   // remove_wall(0, 0, 0) (roomgrid.py:279-311)
        if (g.lane >= 1 && g.lane < rg.rs - 1) g.p[g.lane * g.W + rg.st] = (uint8_t)CELL_EMPTY;
        MG_WAVE_LDS_SYNC();
        } else {
          for (int y = 1; y < rg.rs - 1; y++) g.p[y * g.W + rg.st] = (uint8_t)CELL_EMPTY;
        }
        a = rand_int(rng, 0, per);
        b = per + rand_int(rng, 0, per);
        if (rand_int(rng, 0, 2) == 0) { const int t = a; a = b; b = t; }
      }
      if (rng.dead()) continue;
      const int pa = (int)(opos >> (8 * a)) & 255, pb = (int)(opos >> (8 * b)) & 255;
      if (abs((pa & 15) - (pb & 15)) + abs((pa >> 4) - (pb >> 4)) == 1) continue;      // validate_instrs: "objs already next to each other"
      out.mission = ((uint32_t)(okind >> (8 * a)) & 31u) * 18u + ((uint32_t)(okind >> (8 * b)) & 31u);
      if (P.kind == KIND_PUTNEXT && P.num_crossings) {        // PutNext.reset (:205-214): the object to move starts in the agent's hands;
        out.carry = g.get(pa & 15, pa >> 4);                  // the reset observation still shows it on the grid (FLAG_SHOW_TAKEN)
        g.set(pa & 15, pa >> 4, CELL_EMPTY);
        out.aux = (uint64_t)((pa >> 4) * g.W + (pa & 15));
      }
      return;
    }
    if (P.kind == KIND_ACTIONOBJDOOR) {
      uint64_t ocol = 0, otyp = 0;
      uint32_t used = 0;
      int n = 0;
      while (n < 5 && rg.ok && !rng.dead()) {                 // add_distractors(1, 1, 5): unique
        const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
        if ((used >> (c2 * 3 + t2)) & 1u) continue;
        rg.add_object(rng, g, 1, 1, t2, c2, ti, ci);
        used |= 1u << (c2 * 3 + t2);
        ocol |= (uint64_t)c2 << (3 * n); otyp |= (uint64_t)t2 << (2 * n); n++;
      }
      if (!rg.ok || rng.dead()) continue;
      for (int d = 0; d < 4 && !rng.dead(); d++) {
        const int c2 = rg.add_door(rng, g, 1, 1, -1, -1, 0, dx, dy);
        ocol |= (uint64_t)c2 << (3 * n); otyp |= 3ull << (2 * n); n++;
      }
      rg.place_agent_in(rng, g, 1, 1, out);
      if (!rg.ok || rng.dead()) continue;
      const int k = rand_int(rng, 0, n);
      const bool first = rand_int(rng, 0, 2) == 0;
      const uint32_t kc = (uint32_t)(ocol >> (3 * k)) & 7u, kt = (uint32_t)(otyp >> (2 * k)) & 3u;
      uint32_t matches = 0;
      for (int d = 0; d < n; d++) matches += (((uint32_t)(ocol >> (3 * d)) & 7u) == kc && ((uint32_t)(otyp >> (2 * d)) & 3u) == kt) ? 1u : 0u;
      const uint32_t verb = first ? 0u : (kt == 3u ? 2u : 1u);
      out.mission = verb * 48u + (matches > 1u ? 24u : 0u) + kc * 4u + kt;
      return;
    }
    // OpenDoor
    uint32_t avail = 0x543210u;                               // _rand_subset(COLOR_NAMES, 4)
    int colors[4], doorx[4], doory[4];
    for (int n = 0, na = 6; n < 4; n++, na--) {
      const int k = rand_int(rng, 0, na);
      colors[n] = (int)((avail >> (4 * k)) & 15u);
      const uint32_t lowmask = (1u << (4 * k)) - 1u;
      avail = (avail & lowmask) | ((avail >> 4) & ~lowmask);
    }
    for (int d = 0; d < 4; d++) rg.add_door(rng, g, 1, 1, d, colors[d], 0, doorx[d], doory[d]);
    const bool by_loc = P.num_crossings == 2 || (P.num_crossings == 0 && rand_int(rng, 0, 2) == 1);      // _rand_elem(["color", "loc"])
    const int loc = by_loc ? rand_int(rng, 0, 4) : -1;        // LOC_NAMES = left, right, front, behind
    rg.place_agent_in(rng, g, 1, 1, out);
    if (!rg.ok || rng.dead()) continue;
    uint32_t mask = 0, count = 0;
    for (int d = 0; d < 4; d++) {
      bool match;
      if (!by_loc) match = d == 0;
      else {                                                  // ObjDesc.find_matching_objs with a location (verifier.py:139-160)
        const int vx = doorx[d] - rg.ax, vy = doory[d] - rg.ay, d1x = dir_dx(out.dir), d1y = dir_dy(out.dir), d2x = -d1y, d2y = d1x;
        const int side = vx * d2x + vy * d2y, ahead = vx * d1x + vy * d1y;
        match = loc == 0 ? side < 0 : loc == 1 ? side > 0 : loc == 2 ? ahead > 0 : ahead < 0;
      }
      if (match) { mask |= 1u << color_from_sorted((uint32_t)colors[d]); count++; }
    }
    out.aux = (uint64_t)mask;
    out.mission = by_loc ? 6u + (count > 1u ? 4u : 0u) + (uint32_t)loc : (uint32_t)colors[0];
    return;
  }
  out.failed = true;
}

// ======================================================================================================
// Sentence levels: the BabyAI levels whose mission is an instruction TREE (verifier.py And / Before / After) and / or whose
// descriptions need object identity (locations).  The generator leaves, next to the map, the instruction record of mg_device.h
// (INSTR_WORDS u64 in LDS at `iw`); k_verify evaluates it after every step.
// ======================================================================================================
enum : int { KIND_OPENTWODOORS = 50, KIND_OPENDOORSORDER = 51, KIND_MOVETWOACROSS = 52, KIND_LEVELGEN = 53 };
// lane l = object id l: where it is and what it is.  Ids are given in cell order to everything that is neither None nor a wall
// (RoomGridLevel gives WorldObj identity to the same set; the order is internal).
struct Objs { uint32_t pos, code; int x, y; uint32_t count; };
MG_D Objs assign_ids(GridRef& g, uint64_t* iw) {
  uint16_t* pos = (uint16_t*)(iw + IW_POS);
  MG_WAVE_LDS_SYNC();
  pos[g.lane] = (uint16_t)POS_GONE;
  MG_WAVE_LDS_SYNC();
  uint32_t count = 0;
  const int cells = g.W * g.H;
  for (int base = 0; base < cells; base += 64) {
    const int c = base + g.lane;
    const uint32_t code = c < cells ? (uint32_t)g.p[c] : 0u;
    const bool isobj = c < cells && code != CELL_EMPTY && cell_type(code) != T_WALL;
    const unsigned long long m = __ballot(isobj);
    const uint32_t id = count + (uint32_t)__popcll(m & ((1ull << g.lane) - 1ull));
    if (isobj && id < 63u) pos[id] = (uint16_t)c;
    count += (uint32_t)__popcll(m);
  }
  MG_WAVE_LDS_SYNC();
  Objs o;
  o.pos = pos[g.lane]; o.count = count;
  o.code = o.pos < POS_GONE ? (uint32_t)g.p[o.pos] : 0u;
  o.y = o.pos < POS_GONE ? (int)o.pos / g.W : -9; o.x = o.pos < POS_GONE ? (int)o.pos - o.y * g.W : -9;
  return o;
}
// (one lane: the same ids -- cell order -- in the record's table; returns the object count.  Per-lane forms of the sentence generators: pinned on
// the CPU against the oracle, tests/test_generators_cpu.py; not yet used on the device)
MG_HD uint32_t assign_ids_lane(const LaneGrid& g, uint64_t* iw) {
  uint16_t* pos = (uint16_t*)(iw + IW_POS);
  for (int i = 0; i < 64; i++) pos[i] = (uint16_t)POS_GONE;
  uint32_t count = 0;
  const int cells = g.W * g.H;
  for (int c = 0; c < cells; c++) {
    const uint32_t code = (uint32_t)g.p[c];
    if (code == CELL_EMPTY || cell_type(code) == T_WALL) continue;
    if (count < 63u) pos[count] = (uint16_t)c;
    count++;
  }
  return count;
}
MG_HD uint64_t desc_set_lane(const LaneGrid& g, const uint64_t* iw, uint32_t type, uint32_t colp1, uint32_t loc, int ax, int ay, uint32_t dir, int rs) {
  const uint16_t* pos = (const uint16_t*)(iw + IW_POS);
  uint64_t set = 0;
  for (int i = 0; i < 63; i++) {
    const uint32_t q = pos[i];
    if (q >= POS_GONE) continue;
    const uint32_t code = (uint32_t)g.p[q];
    const int oy = (int)q / g.W, ox = (int)q - oy * g.W;
    bool m = cell_ref_type(code) == type && (colp1 == 0u || cell_color(code) == colp1 - 1u);
    if (loc) {
      const int st = rs - 1, top_x = (ax / st) * st, top_y = (ay / st) * st;
      const bool inside = ox >= top_x && ox < top_x + rs && oy >= top_y && oy < top_y + rs;
      const int vx = ox - ax, vy = oy - ay, d1x = dir_dx(dir), d1y = dir_dy(dir), d2x = -d1y, d2y = d1x;
      const int side = vx * d2x + vy * d2y, ahead = vx * d1x + vy * d1y;
      m = m && inside && (loc == 1u ? side < 0 : loc == 2u ? side > 0 : loc == 3u ? ahead > 0 : ahead < 0);
    }
    set |= (uint64_t)m << i;
  }
  return set;
}
// ObjDesc.find_matching_objs(use_location=True) (verifier.py:105-171) as one ballot over the object ids
MG_D uint64_t desc_set(const Objs& o, uint32_t type, uint32_t colp1, uint32_t loc, int ax, int ay, uint32_t dir, int rs) {
  bool m = o.pos < POS_GONE && cell_ref_type(o.code) == type && (colp1 == 0u || cell_color(o.code) == colp1 - 1u);
  if (loc) {
    const int st = rs - 1, top_x = (ax / st) * st, top_y = (ay / st) * st;                       // room_from_pos(agent_pos).pos_inside
    const bool inside = o.x >= top_x && o.x < top_x + rs && o.y >= top_y && o.y < top_y + rs;
    const int vx = o.x - ax, vy = o.y - ay, d1x = dir_dx(dir), d1y = dir_dy(dir), d2x = -d1y, d2y = d1x;
    const int side = vx * d2x + vy * d2y, ahead = vx * d1x + vy * d1y;
    m = m && inside && (loc == 1u ? side < 0 : loc == 2u ? side > 0 : loc == 3u ? ahead > 0 : ahead < 0);
  }
  return __ballot(m);
}
// the record's constant parts once the tree is known: header, stale sets, mission words
template <class G>
MG_HD void sentence_finish(G& g, uint64_t* iw, uint32_t root, const uint32_t node[3], uint32_t max_steps) {
  if (!G::kWave || g.lane == 0) {
    uint64_t h = (uint64_t)root | ((uint64_t)max_steps << 39);
    for (int n = 0; n < 3; n++) h |= (uint64_t)node[n] << (3 + 8 * n);
    iw[0] = h;
    for (int k = 0; k < 8; k++) iw[IW_STALE + k] = ~0ull;
    uint64_t m0 = (uint64_t)root << 60, m1 = 0;
    for (int k = 0; k < 3; k++) m0 |= (iw[IW_LEAF + k] & 0xFFFFFull) << (20 * k);
    m1 = (iw[IW_LEAF + 3] & 0xFFFFFull) | ((uint64_t)node[0] << 20) | ((uint64_t)node[1] << 28) | ((uint64_t)node[2] << 36);
    iw[IW_MISSION] = m0; iw[IW_MISSION + 1] = m1; iw[IW_MISSION + 2] = 0ull;
  }
  if constexpr (G::kWave) MG_WAVE_LDS_SYNC();
}
template <class G>
MG_HD void sentence_leaf(G& g, uint64_t* iw, int k, uint32_t verb, uint32_t d9, uint64_t dset, uint32_t f9, uint64_t fset, uint32_t strict) {
  if (!G::kWave || g.lane == 0) {
    const uint32_t da = d9 | ((__builtin_popcountll(dset) > 1 ? 1u : 0u) << 8), fa = f9 | ((__builtin_popcountll(fset) > 1 ? 1u : 0u) << 8);
    iw[IW_LEAF + k] = (uint64_t)leaf20(verb, da, fa) | ((uint64_t)strict << 20);
    iw[IW_SET + 2 * k] = dset; iw[IW_SET + 2 * k + 1] = fset;
  }
}

// sequences of two instructions: OpenTwoDoors (open.py:306-325; P.start_x / P.start_y = first / second colour as a COLOR_NAMES index
// or -1, P.strip2_row = strict), OpenDoorsOrder (:399-425; P.num_dists = num_doors, P.strip2_row = debug), MoveTwoAcross
// (other.py:404-428; P.num_dists = objs_per_room).  max_steps is the class's fixed value (P.max_steps).
template <class R, class G>
MG_HD void gen_babyai_seq(R& rng, G& g, const GenParams& P, GenResult& out, uint64_t* iw) {
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    rng.checkpoint();
    RG rg;
    rg.gen_grid(rng, g, P.room_size);
    int dx, dy;
    out.aux = ~0ull; out.carry = 0; out.mission = 0;
    uint32_t node[3] = { 0u, 0u, 0u }, root = 0;
    if constexpr (G::kWave) { if (g.lane < INSTR_WORDS) iw[g.lane] = 0ull; }
    else for (int k = 0; k < INSTR_WORDS; k++) iw[k] = 0ull;
    if (P.kind == KIND_OPENTWODOORS || P.kind == KIND_OPENDOORSORDER) {
      const int nsub = P.kind == KIND_OPENTWODOORS ? 2 : P.num_dists;
      uint32_t avail = 0x543210u;                             // _rand_subset(COLOR_NAMES, n)
      int colors[4] = { 0, 0, 0, 0 };
      for (int n = 0, na = 6; n < nsub; n++, na--) {
        const int k = rand_int(rng, 0, na);
        const int c = (int)((avail >> (4 * k)) & 15u);
        if (n == 0) colors[0] = c; else if (n == 1) colors[1] = c; else if (n == 2) colors[2] = c; else colors[3] = c;
        const uint32_t lowmask = (1u << (4 * k)) - 1u;
        avail = (avail & lowmask) | ((avail >> 4) & ~lowmask);
      }
      int c1, c2;
      uint32_t strict2 = 0;
      if (P.kind == KIND_OPENTWODOORS) {
        c1 = P.start_x >= 0 ? P.start_x : colors[0]; c2 = P.start_y >= 0 ? P.start_y : colors[1];
        rg.add_door(rng, g, 1, 1, 2, c1, 0, dx, dy);
        rg.add_door(rng, g, 1, 1, 0, c2, 0, dx, dy);
        rg.place_agent_in(rng, g, 1, 1, out);
        if (!rg.ok || rng.dead()) continue;
        node[0] = node8(N_BEFORE, 0, 1); root = 4;
      } else {
        for (int d = 0; d < nsub && !rng.dead(); d++) {
          const int c = d == 0 ? colors[0] : d == 1 ? colors[1] : d == 2 ? colors[2] : colors[3];
          rg.add_door(rng, g, 1, 1, -1, c, 0, dx, dy);
        }
        rg.place_agent_in(rng, g, 1, 1, out);
        if (!rg.ok || rng.dead()) continue;
        const int a = rand_int(rng, 0, nsub);                 // _rand_subset(doors, 2)
        int b = rand_int(rng, 0, nsub - 1); if (b >= a) b++;
        const int mode = rand_int(rng, 0, 3);
        c1 = a == 0 ? colors[0] : a == 1 ? colors[1] : a == 2 ? colors[2] : colors[3];
        c2 = b == 0 ? colors[0] : b == 1 ? colors[1] : b == 2 ? colors[2] : colors[3];
        strict2 = (uint32_t)P.strip2_row;
        if (mode == 0) root = 0; else { node[0] = node8(mode == 1 ? N_BEFORE : N_AFTER, 0, 1); root = 4; }
      }
      if (rng.dead()) continue;
      const uint32_t k1 = color_from_sorted((uint32_t)c1) + 1u, k2 = color_from_sorted((uint32_t)c2) + 1u;
      uint64_t s1, s2;
      if constexpr (G::kWave) {
      const Objs o = assign_ids(g, iw);
      s1 = desc_set(o, T_DOOR, k1, 0, rg.ax, rg.ay, out.dir, rg.rs); s2 = desc_set(o, T_DOOR, k2, 0, rg.ax, rg.ay, out.dir, rg.rs);
      } else {
        assign_ids_lane(g, iw);
        s1 = desc_set_lane(g, iw, T_DOOR, k1, 0, rg.ax, rg.ay, out.dir, rg.rs); s2 = desc_set_lane(g, iw, T_DOOR, k2, 0, rg.ax, rg.ay, out.dir, rg.rs);
      }
      sentence_leaf(g, iw, 0, V_OPEN, desc9(T_DOOR, k1, 0, 0), s1, 0, 0, (uint32_t)P.strip2_row);
      sentence_leaf(g, iw, 1, V_OPEN, desc9(T_DOOR, k2, 0, 0), s2, 0, 0, strict2);
      sentence_finish(g, iw, root, node, (uint32_t)P.max_steps);
      return;
    }
    // MoveTwoAcross
    const int per = P.num_dists;
    rg.place_agent_in(rng, g, 0, 0, out);
    uint64_t opos[3] = { 0, 0, 0 }, okind[3] = { 0, 0, 0 };   // byte n & 7 of word n >> 3: x | y << 4; colour * 3 + type (up to 18 objects)
    uint32_t used = 0;
    int n = 0;
    for (int room = 0; room < 2 && rg.ok && !rng.dead(); room++)
      for (int k = 0; k < per && rg.ok && !rng.dead();) {
        const int c2 = rand_int(rng, 0, 6), t2 = rand_int(rng, 0, 3);
        if ((used >> (c2 * 3 + t2)) & 1u) continue;
        int x, y;
        if (!place_obj(rng, g, make_cell((uint32_t)T_KEY + (uint32_t)t2, color_from_sorted((uint32_t)c2)), room * rg.st, 0, rg.rs, rg.rs, rg.ax, rg.ay, true, 1000, x, y)) { rg.ok = false; break; }
        used |= 1u << (c2 * 3 + t2);
        const int w = n >> 3, sh = 8 * (n & 7);
        const uint64_t pv = (uint64_t)(x | (y << 4)) << sh, kv = (uint64_t)(c2 * 3 + t2) << sh;
        if (w == 0) { opos[0] |= pv; okind[0] |= kv; } else if (w == 1) { opos[1] |= pv; okind[1] |= kv; } else { opos[2] |= pv; okind[2] |= kv; }
        n++; k++;
      }
    if (!rg.ok || rng.dead()) continue;
    if constexpr (G::kWave) {
    MG_WAVE_LDS_SYNC();                                       // remove_wall(0, 0, 0)
    if (g.lane >= 1 && g.lane < rg.rs - 1) g.p[g.lane * g.W + rg.st] = (uint8_t)CELL_EMPTY;
    MG_WAVE_LDS_SYNC();
    } else {
      for (int y = 1; y < rg.rs - 1; y++) g.p[y * g.W + rg.st] = (uint8_t)CELL_EMPTY;
    }
    const int l0 = rand_int(rng, 0, per); int l1 = rand_int(rng, 0, per - 1); if (l1 >= l0) l1++;       // _rand_subset(objs_l, 2)
    const int r0 = rand_int(rng, 0, per); int r1 = rand_int(rng, 0, per - 1); if (r1 >= r0) r1++;
    if (rng.dead()) continue;
    const int ia = l0, ib = per + r0, ic = per + r1, id = l1;
    auto pos_of = [&](int i) -> int { const uint64_t w = (i >> 3) == 0 ? opos[0] : (i >> 3) == 1 ? opos[1] : opos[2]; return (int)(w >> (8 * (i & 7))) & 255; };
    auto kind_of = [&](int i) -> uint32_t { const uint64_t w = (i >> 3) == 0 ? okind[0] : (i >> 3) == 1 ? okind[1] : okind[2]; return (uint32_t)(w >> (8 * (i & 7))) & 31u; };
    auto adjacent = [&](int i, int j) -> bool { const int p = pos_of(i), q = pos_of(j); return abs((p & 15) - (q & 15)) + abs((p >> 4) - (q >> 4)) == 1; };
    if (adjacent(ia, ib) || adjacent(ic, id)) continue;       // validate_instrs: "objs already next to each other" (the objects are unique)
    auto d9_of = [&](int i) -> uint32_t { const uint32_t kd = kind_of(i); return desc9((uint32_t)T_KEY + kd % 3u, color_from_sorted(kd / 3u) + 1u, 0, 0); };
    uint64_t sa, sb, sc, sd;
    if constexpr (G::kWave) {
    const Objs o = assign_ids(g, iw);
    auto set_of = [&](int i) -> uint64_t { const uint32_t kd = kind_of(i); return desc_set(o, (uint32_t)T_KEY + kd % 3u, color_from_sorted(kd / 3u) + 1u, 0, rg.ax, rg.ay, out.dir, rg.rs); };
    sa = set_of(ia); sb = set_of(ib); sc = set_of(ic); sd = set_of(id);
    } else {
      assign_ids_lane(g, iw);
      auto set_of = [&](int i) -> uint64_t { const uint32_t kd = kind_of(i); return desc_set_lane(g, iw, (uint32_t)T_KEY + kd % 3u, color_from_sorted(kd / 3u) + 1u, 0, rg.ax, rg.ay, out.dir, rg.rs); };
      sa = set_of(ia); sb = set_of(ib); sc = set_of(ic); sd = set_of(id);
    }
    sentence_leaf(g, iw, 0, V_PUTNEXT, d9_of(ia), sa, d9_of(ib), sb, 0);
    sentence_leaf(g, iw, 1, V_PUTNEXT, d9_of(ic), sc, d9_of(id), sd, 0);
    node[0] = node8(N_BEFORE, 0, 1); root = 4;
    sentence_finish(g, iw, root, node, (uint32_t)P.max_steps);
    return;
  }
  out.failed = true;
}

// envs/babyai/core/levelgen.py: LevelGen (PickupLoc, GoToSeq, Synth*, MiniBossLevel, BossLevel*).  P.num_crossings = action kinds
// (bit 0 goto, 1 pickup, 2 open, 3 putnext) | instr kinds (bit 4 action, 5 and, 6 seq) | bit 7 locations | bit 8 unblocking |
// bit 9 implicit_unlock; P.strip2_row = locked_room_prob in percent; P.num_dists distractors.  LevelGen.locked_room survives from
// episode to episode (only __init__ clears it), and rand_obj looks at it: `st[0]` = (i | j << 4 | 0x100 valid) as the env's previous
// episode left it, st[1] = as of the current attempt's checkpoint; out.gstate = as this episode leaves it.
template <class R, class G>
MG_HD void gen_levelgen(R& rng, G& g, const GenParams& P, GenResult& out, uint64_t* iw, uint32_t* st) {
  const int flags = P.num_crossings;
  uint32_t locked;
  if constexpr (G::kWave) {
  MG_WAVE_LDS_SYNC();
  locked = uni32(out.resume ? st[1] : st[0]);
  } else locked = out.resume ? st[1] : st[0];
  for (uint32_t attempt = 0; attempt < 4096 && !rng.dead(); attempt++) {
    out.retries = attempt;
    MG_GA(out, 6);                              // (what the previous attempt spent since its last mark: instruction drawing + validation)
    MG_GA_ATTEMPT(out);
    rng.checkpoint();
    if (!G::kWave || g.lane == 0) st[1] = locked;
    bool fresh = false;
    RG rg;
    rg.gen_grid(rng, g, P.room_size);
    MG_GA(out, 1);
    out.aux = ~0ull; out.carry = 0; out.mission = 0; out.gstate = locked;
    if constexpr (G::kWave) { if (g.lane < INSTR_WORDS) iw[g.lane] = 0ull; }
    else for (int k = 0; k < INSTR_WORDS; k++) iw[k] = 0ull;
    int dx, dy, ti, ci;
    // _rand_float(0, 1) = Generator.uniform: next_double = (next_uint64 >> 11) / 2^53
    const double u = (double)(rng.next64() >> 11) * (1.0 / 9007199254740992.0);
    if (u < (double)P.strip2_row / 100.0) {                   // add_locked_room (levelgen.py:82-111)
      int dc = 0;
      while (!rng.dead()) {
        const int i = rand_int(rng, 0, rg.nc), j = rand_int(rng, 0, rg.nr), k = rand_int(rng, 0, 4);
        locked = (uint32_t)i | ((uint32_t)j << 4) | 0x100u; fresh = true;
        if (!rg.has_nb(i, j, k)) continue;
        dc = rg.add_door(rng, g, i, j, k, -1, 1, dx, dy);
        break;
      }
      while (!rng.dead()) {
        const int i = rand_int(rng, 0, rg.nc), j = rand_int(rng, 0, rg.nr);
        if (i == (int)(locked & 15u) && j == (int)((locked >> 4) & 15u)) continue;
        rg.add_object(rng, g, i, j, 0, dc, ti, ci);
        break;
      }
    }
    MG_GA(out, 4);
    if (!rg.ok || rng.dead()) continue;
    rg.connect_all(rng, g, -1);
    MG_GA(out, 3);
    if (!rg.ok || rng.dead()) continue;
#pragma unroll 1
    for (int n = 0; n < P.num_dists && rg.ok && !rng.dead(); n++) {     // add_distractors(all_unique=False) over random rooms
      int c2, t2, ri, rj, v4[4];
      bool four = false;
      if constexpr (G::kWave) four = draw4_spec(rng, g, 6u, 3u, (uint32_t)rg.nc, (uint32_t)rg.nr, v4);
      if (four) { c2 = v4[0]; t2 = v4[1]; ri = v4[2]; rj = v4[3]; }
      else { c2 = rand_int(rng, 0, 6); t2 = rand_int(rng, 0, 3); ri = rand_int(rng, 0, rg.nc); rj = rand_int(rng, 0, rg.nr); }
      rg.add_object(rng, g, ri, rj, t2, c2, ti, ci);
    }
    MG_GA(out, 4);
    if (!rg.ok || rng.dead()) continue;
    while (!rng.dead()) {
      const int ai = rand_int(rng, 0, rg.nc), aj = rand_int(rng, 0, rg.nr);
      rg.place_agent_in(rng, g, ai, aj, out);
      if (!rg.ok || rng.dead()) break;
      if (fresh && rg.ax / rg.st == (int)(locked & 15u) && rg.ay / rg.st == (int)((locked >> 4) & 15u)) continue;
      break;
    }
    MG_GA(out, 2);
    if (!rg.ok || rng.dead()) continue;
    const bool reachable = ((flags >> 8) & 1) || maze_objs_reachable(g, rg.ax, rg.ay);
    MG_GA(out, 5);
    if (!reachable) continue;
    Objs o;
    if constexpr (G::kWave) o = assign_ids(g, iw);
    else { o.pos = 0; o.code = 0; o.x = 0; o.y = 0; o.count = assign_ids_lane(g, iw); }
    if (o.count > 63u) { out.failed = true; return; }
    // rand_instr (levelgen.py:157-211) without the recursion: root = action | And(action, action) | Before / After(sub, sub), sub =
    // action | And(action, action).  rand_obj (:113-155) = draw (colour | None, type, location?) until something matches.
    bool fail = false;
    uint32_t nleaf = 0, nnode = 0, navs = 0;
    uint32_t node[3] = { 0u, 0u, 0u };
    uint64_t lastset = 0;
    auto rand_obj = [&](int types) -> uint32_t {              // types: 0 OBJ_TYPES, 1 OBJ_TYPES_NOT_DOOR, 2 ["door"]; returns desc9, set in lastset
      for (int tries = 0; !rng.dead(); ) {
        if (tries > 100) { fail = true; return 0u; }          // RecursionError("failed to find suitable object")
        tries++;
        const int c = rand_int(rng, 0, 7);                    // _rand_elem([None, *COLOR_NAMES])
        const int tk = types == 2 ? 3 : rand_int(rng, 0, types == 1 ? 3 : 4);       // OBJ_TYPES = [box, ball, key, door]
        const uint32_t t = tk == 0 ? (uint32_t)T_BOX : tk == 1 ? (uint32_t)T_BALL : tk == 2 ? (uint32_t)T_KEY : (uint32_t)T_DOOR;
        uint32_t loc = 0;
        if ((flags >> 7) & 1) if (rand_int(rng, 0, 2) == 0) loc = 1u + (uint32_t)rand_int(rng, 0, 4);
        const uint32_t colp1 = c ? color_from_sorted((uint32_t)(c - 1)) + 1u : 0u;
        uint64_t set;
        if constexpr (G::kWave) set = desc_set(o, t, colp1, loc, rg.ax, rg.ay, out.dir, rg.rs);
        else set = desc_set_lane(g, iw, t, colp1, loc, rg.ax, rg.ay, out.dir, rg.rs);
        if (set == 0ull) continue;
        if (!((flags >> 9) & 1) && (locked & 0x100u)) {       // isinstance(self.locked_room, Room): possibly last episode's
          const int tx = (int)(locked & 15u) * rg.st, ty = (int)((locked >> 4) & 15u) * rg.st;
          if constexpr (G::kWave) {
          const bool outside = ((set >> g.lane) & 1ull) && !(o.x >= tx && o.x < tx + rg.rs && o.y >= ty && o.y < ty + rg.rs);
          if (__ballot(outside) == 0ull) continue;
          } else {
            bool any_outside = false;
            const uint16_t* pos = (const uint16_t*)(iw + IW_POS);
            for (int i = 0; i < 63; i++)
              if ((set >> i) & 1ull) { const int oy = (int)pos[i] / g.W, ox = (int)pos[i] - oy * g.W; any_outside |= !(ox >= tx && ox < tx + rg.rs && oy >= ty && oy < ty + rg.rs); }
            if (!any_outside) continue;
          }
        }
        lastset = set;
        return desc9(t, colp1, loc, 0);
      }
      fail = true; return 0u;
    };
    auto action = [&]() -> uint32_t {                         // "action": one of the enabled verbs about random description(s)
      uint32_t acts = 0; int na = 0;
      for (int k = 0; k < 4; k++) if ((flags >> k) & 1) { acts |= (uint32_t)k << (2 * na); na++; }
      const uint32_t verb = (acts >> (2 * rand_int(rng, 0, na))) & 3u;
      uint32_t d9 = 0, f9 = 0; uint64_t ds = 0, fs = 0;
      if (verb == V_GOTO) { d9 = rand_obj(0); ds = lastset; }
      else if (verb == V_PICKUP) { d9 = rand_obj(1); ds = lastset; }
      else if (verb == V_OPEN) { d9 = rand_obj(2); ds = lastset; }
      else { d9 = rand_obj(1); ds = lastset; if (!fail) { f9 = rand_obj(0); fs = lastset; } }
      if (fail || nleaf >= 4u) { fail = true; return 0u; }
      sentence_leaf(g, iw, (int)nleaf, verb, d9, ds, f9, fs, 0);
      navs += verb == V_PUTNEXT ? 2u : 1u;
      return nleaf++;
    };
    auto and_node = [&]() -> uint32_t {
      const uint32_t a = action();
      const uint32_t b = fail ? 0u : action();
      if (fail || nnode >= 3u) { fail = true; return 0u; }
      node[nnode] = node8(N_AND, a, b);
      return 4u + nnode++;
    };
    auto pick = [&](int kinds) -> int {                       // _rand_elem of the enabled instruction kinds (0 action, 1 and, 2 seq)
      uint32_t list = 0; int nk = 0;
      for (int k = 0; k < 3; k++) if ((kinds >> k) & 1) { list |= (uint32_t)k << (2 * nk); nk++; }
      return (int)((list >> (2 * rand_int(rng, 0, nk))) & 3u);
    };
    uint32_t root;
    {
      const int kind = pick((flags >> 4) & 7);
      if (kind == 0) root = action();
      else if (kind == 1) root = and_node();
      else {
        const uint32_t a = pick(3) == 0 ? action() : and_node();
        uint32_t b = 0;
        if (!fail) b = pick(3) == 0 ? action() : and_node();
        if (fail || nnode >= 3u) { fail = true; root = 0; }
        else {
          node[nnode] = node8(rand_int(rng, 0, 2) == 0 ? N_BEFORE : N_AFTER, a, b);
          root = 4u + nnode++;
        }
      }
    }
    if (fail || rng.dead()) continue;
    // validate_instrs (roomgrid_level.py:146-203)
    bool reject = false;
    uint32_t locked_colors = 0;
    if constexpr (G::kWave) {
    MG_WAVE_LDS_SYNC();
    if ((flags >> 8) & 1) {
      for (int base = 0; base < g.W * g.H; base += 64) {
        const int q = base + g.lane;
        const uint32_t v = q < g.W * g.H ? (uint32_t)g.p[q] : 0u;
        const bool lk = q < g.W * g.H && cell_type(v) == T_DOOR_LOCKED;
        for (uint32_t c = 0; c < 6u; c++) if (__ballot(lk && cell_color(v) == c)) locked_colors |= 1u << c;
      }
    }
    } else if ((flags >> 8) & 1) {
      for (int q = 0; q < g.W * g.H; q++) { const uint32_t v = (uint32_t)g.p[q]; if (cell_type(v) == T_DOOR_LOCKED) locked_colors |= 1u << cell_color(v); }
    }
    for (uint32_t k = 0; k < nleaf; k++) {
      uint32_t l20; uint64_t ds, fs;
      if constexpr (G::kWave) { l20 = (uint32_t)uni64(iw[IW_LEAF + k]) & 0xFFFFFu; ds = uni64(iw[IW_SET + 2 * k]); fs = uni64(iw[IW_SET + 2 * k + 1]); }
      else { l20 = (uint32_t)iw[IW_LEAF + k] & 0xFFFFFu; ds = iw[IW_SET + 2 * k]; fs = iw[IW_SET + 2 * k + 1]; }
      const uint32_t verb = l20 & 3u, d9 = (l20 >> 2) & 511u, f9 = (l20 >> 11) & 511u;
      if (verb == V_PUTNEXT) {
        if (ds & fs) reject = true;                           // "there are objects that match both lhs and rhs of PutNext"
        if constexpr (G::kWave) {
        bool next = false;                                    // objs_next(): some object to move already lies next to a fixed one
        const bool mine = (ds >> g.lane) & 1ull;
        for (int m = 0; m < 63; m++)
          if ((fs >> m) & 1ull) {
            const int fxm = (int)lane32((uint32_t)o.x, (uint32_t)m), fym = (int)lane32((uint32_t)o.y, (uint32_t)m);
            next |= mine && abs(o.x - fxm) + abs(o.y - fym) == 1;
          }
        if (__ballot(next)) reject = true;
        } else {
          const uint16_t* pos = (const uint16_t*)(iw + IW_POS);
          for (int a_ = 0; a_ < 63; a_++)
            if ((ds >> a_) & 1ull)
              for (int m = 0; m < 63; m++)
                if ((fs >> m) & 1ull) {
                  const int ay_ = (int)pos[a_] / g.W, ax_ = (int)pos[a_] - ay_ * g.W, my = (int)pos[m] / g.W, mx = (int)pos[m] - my * g.W;
                  if (abs(ax_ - mx) + abs(ay_ - my) == 1) reject = true;
                }
        }
      }
      if ((flags >> 8) & 1) {                                 // unblocking: "cannot do anything with/to a locked door's key"
        if (desc9_type(d9) == T_KEY && desc9_color(d9) && ((locked_colors >> (desc9_color(d9) - 1u)) & 1u)) reject = true;
        if (verb == V_PUTNEXT && desc9_type(f9) == T_KEY && desc9_color(f9) && ((locked_colors >> (desc9_color(f9) - 1u)) & 1u)) reject = true;
      }
    }
    if (reject) continue;
    // RoomGridLevel.reset (roomgrid_level.py:71-85): max_steps = num_navs_needed * room_size**2 * num_rows * num_cols
    sentence_finish(g, iw, root, node, navs * (uint32_t)(rg.rs * rg.rs * rg.nc * rg.nr));
    out.gstate = locked;
    MG_GA(out, 6);
    return;
  }
  out.gstate = locked;
  out.failed = true;
}

// Groups.  The generator kernels (k_generate / k_refill) are instantiated per generator group, so that a level's generator
// launch only carries -- and only pays registers for -- the generators its env kind can need: with every kind inlined into one
// kernel the register count is that of the largest one (LevelGen), which costs the BASELINE levels occupancy (GoToRedBall's
// refill: +14 %).  k_step is instantiated per RULE group with the same constants (mg_api.hip rule_group).
//   GG_LIGHT single-room levels (+ DynamicObstacles)   GG_ROOMGRID RoomGrid-based levels (incl. GoToRedBall) + GoToObject
//   GG_ROOMS the multi-room MiniGrid and BabyAI levels with one instruction (kinds 21..49)   GG_SENTENCE kinds 50..53
// (the GG_* constants and gen_group_of_kind live in mg_device.h: the step kernels use them without the generators)

template <int GG, class R>
MG_D void generate_episode(R& rng, GridRef& g, const GenParams& P, GenResult& out) {
  out.ax = out.ay = 1; out.dir = 0; out.mission = 0; out.retries = 0; out.failed = false; out.aux = 0;
  if constexpr (GG == GG_LIGHT || GG == GG_ALL) {
    switch (P.kind) {
      case 0: gen_empty(rng, g, P, out); return;
      case 1: gen_doorkey(rng, g, P, out); return;
      case 2: gen_crossing(rng, g, P, out); return;
      case 4: gen_lavagap(rng, g, P, out); return;
      case 5: gen_distshift(rng, g, P, out); return;
      case 6: gen_fourrooms(rng, g, P, out); return;
      case 7: gen_fetch(rng, g, P, out); return;
      case 8: gen_gotodoor(rng, g, P, out); return;
      case 12: gen_redbluedoors(rng, g, P, out); return;
      case 13: gen_memory(rng, g, P, out); return;
      case 15: gen_dynobs(rng, g, P, out); return;
      default: break;
    }
  }
  if constexpr (GG == GG_ROOMGRID || GG == GG_ALL) {
    switch (P.kind) {
      case 3: case 16: case 17: case 18: case 19: gen_goto(rng, g, P, out); return;
      case 9: gen_unlock_family(rng, g, P, out, 0); return;
      case 10: gen_unlock_family(rng, g, P, out, 1); return;
      case 11: gen_unlock_family(rng, g, P, out, 2); return;
      case 14: gen_keycorridor(rng, g, P, out); return;
      case 20: gen_gotoobject(rng, g, P, out); return;
      default: break;
    }
  }
  if constexpr (GG == GG_ROOMS || GG == GG_ALL) {
    switch (P.kind) {
      case 21: gen_lockedroom(rng, g, P, out); return;
      case 22: gen_playground(rng, g, P, out); return;
      case 23: gen_multiroom(rng, g, P, out); return;
      case 24: case 25: case 27: gen_pickup_level(rng, g, P, out); return;
      case 26: gen_openreddoor(rng, g, P, out); return;
      case 28: gen_findobj(rng, g, P, out); return;
      case 29: gen_unlocklocal(rng, g, P, out); return;
      case 30: gen_keycorridor(rng, g, P, out); out.mission = 2u; return;     // BabyAI KeyCorridor (other.py:252-272): "pick up the ball"
      case 31: gen_obstructedmaze(rng, g, P, out); return;
      case 32: gen_putnear(rng, g, P, out); return;
      case 33: case 34: case 35: gen_babyai_maze(rng, g, P, out); return;
      case 36: case 37: case 38: case 39: case 40: case 41: case 42: case 43: case 44: case 45: gen_babyai_levels(rng, g, P, out); return;
      case 46: case 47: case 48: case 49: gen_babyai_put_open(rng, g, P, out); return;
      default: break;
    }
  }
  if constexpr (GG == GG_SENTENCE || GG == GG_ALL) {
    switch (P.kind) {
      case 50: case 51: case 52: gen_babyai_seq(rng, g, P, out, (uint64_t*)(g.p + P.instr_off)); return;
      case 53: gen_levelgen(rng, g, P, out, (uint64_t*)(g.p + P.instr_off), (uint32_t*)(g.p + P.scratch_off)); return;
      default: break;
    }
  }
  out.failed = true;      // a kind this instantiation was not built for (the host never launches that)
}

}  // namespace mg
