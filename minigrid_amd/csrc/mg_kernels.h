// mg_kernels.h — HIP kernels for gfx950 (MI355X): the lockstep MiniGridEnv.step()/gen_obs()/FullyObs hot path,
// the queue-driven episode generator, and seeding.  One wavefront lane per environment.
//
// HBM layout (all per mg_env handle; N envs, env-major):
//   grid        u8  [N][CS]   one byte per cell (mg_device.h), row-major y*W+x, CS = W*H rounded up to 16
//   spare_grid  u8  [N][CS]   the NEXT episode's map, generated ahead of time (see below)
//   agent       u64 [N]       packed agent record (x, y, dir, carrying, step_count, flags, mission)
//   spare_agent u64 [N]
//   rng / rng_snap u64 [5][N] SoA generator state (current, and as it was before the spare was drawn)
//   obs u8 [N][147 | W*H*3], reward f64 [N], terminated/truncated/direction/mission u8 [N]
//
// Why a spare episode: in the reference an env's np_random stream is consumed ONLY by reset() on this path, so the
// map of episode k+1 can be drawn any time after episode k's map without changing the stream.  The step kernel
// therefore never runs a generator: on (auto)reset it copies the pre-generated spare (CS bytes) and enqueues the
// env id; a separate generator kernel (one wavefront per enqueued env, see mg_gen.h) refills those spares.  The
// sequential PCG64 + rejection-sampling code stays off the step critical path.
#pragma once
#include "mg_device.h"
#include "mg_gen.h"
#include "mg_tiles.h"
#include "mg_rng.h"

namespace mg {

constexpr int VIEW = 7;
constexpr int VIEW_CELLS = VIEW * VIEW;        // 49
constexpr int PARTIAL_OBS_BYTES = VIEW_CELLS * 3;  // 147

enum : int { PHASE_STEP = 0, PHASE_OBSERVE = 1 };
enum : int { ACT_SRC_BUFFER = 0, ACT_SRC_PHILOX = 1 };
enum : int { RULE_NONE = 0, RULE_GOTO = 1, RULE_FETCH = 2, RULE_GOTODOOR = 3, RULE_UNLOCK = 4, RULE_PICKUP = 5,
              RULE_REDBLUE = 6, RULE_MEMORY = 7, RULE_DYNOBS = 8, RULE_GOTOOBJ = 9,
              RULE_PICKUPDESC = 10, RULE_OPENFRONT = 11 };

struct StepParams {
  // state
  uint8_t* grid; const uint8_t* spare_grid; uint64_t* agent; const uint64_t* spare_agent;
  uint64_t* aux; const uint64_t* spare_aux;      // BabyAI GoTo levels: bitboard of the tracked target positions
  // inputs
  const void* actions; int act_dtype; int act_src; uint64_t action_seed; uint32_t t;
  // outputs
  uint8_t* obs; double* reward; uint8_t* term; uint8_t* trunc; uint8_t* dir_out; uint8_t* mission_out;
  // tables / bookkeeping
  const double* reward_lut; uint32_t* refill_queue; uint32_t* refill_count; uint32_t* err;
  unsigned long long* counters;
  // config
  int N, W, H, CS, GS, cells, max_steps, see_through, rule, rule_cell, rule_div, autoreset_next_step, phase, static_gen, gen_blocks;
  int live_gen;           // resets are drawn in place right before the step launch (DynamicObstacles): queue the ended envs
  int off_grid, off_trow, off_vis, off_T, off_lut, off_act, OBE;   // LDS carve-up (bytes); OBE = obs bytes per env
  int view;               // agent view size V (odd, 3..15)
  int no_death_mask; double death_cost;   // NoDeath wrapper (wrappers.py:845-882)
  uint32_t cpe_magic;     // ceil(2^20 / (CS/16))
  int rgb_full, rgb_highlight;   // MODE 4 (tile map for k_render): whole grid + highlight mask instead of the agent's view
  long long env_base;
};

// _reward() = 1 - 0.9 * (step_count / max_steps), three separately rounded f64 ops (minigrid_env.py:240-245).
// Normally read from the host-built LUT; this exact device form covers step_count > max_steps (autoreset disabled).
MG_D double reward_exact(uint32_t step, int max_steps) {
  double q = __ddiv_rn((double)step, (double)max_steps);
  double p = __dmul_rn(0.9, q);
  return __dsub_rn(1.0, p);
}

MG_D uint32_t load_action(const StepParams& P, int e) {
  if (P.act_src == ACT_SRC_PHILOX) {
    uint64_t gi = (uint64_t)(P.env_base + e);
    uint32_t c[4] = { (uint32_t)gi, (uint32_t)(gi >> 32), P.t, 0x41435431u };
    philox4x32_10(c, (uint32_t)P.action_seed, (uint32_t)(P.action_seed >> 32));
    return (uint32_t)(((uint64_t)c[0] * 7u) >> 32);      // uniform over Discrete(7) (minigrid_env.py:63)
  }
  if (P.act_dtype == 0) return ((const uint8_t*)P.actions)[e];
  if (P.act_dtype == 1) return (uint32_t)((const int32_t*)P.actions)[e];
  long long v = ((const long long*)P.actions)[e];
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
// Episode generation (the reference's _gen_grid, see mg_gen.h): one wavefront draws one episode.
// Work list: the refill queue written by k_step, or all envs selected by `mask` (explicit reset(seed=...)).
// ======================================================================================================
struct GenArgs {
  GenParams gp;
  uint8_t* dst_grid; uint64_t* dst_agent;
  uint64_t* rng; uint64_t* rng_snap;                 // rng_snap != null: save the pre-draw state there first
  const uint32_t* queue; const uint32_t* count;      // queue mode (count read on device)
  uint32_t* zero_count;                              // a queue counter nobody uses during this launch: cleared
  const uint8_t* mask;                               // direct mode: optional per-env mask
  uint32_t* err; unsigned long long* counters;
  int N, CS;
  int cap_words;                                     // draw-buffer capacity per generating wave (LDS), in words
  int stat_gen_off;                                  // first generator statistics slot in `counters`
  uint64_t* dst_aux;                                 // auxiliary word of the generated episode (GenResult.aux) or null
  int live;                                          // 1: queue entries are regenerated IN PLACE (dst = live state): only
                                                     //    envs still flagged RESET_PENDING are drawn, and come out FRESH
};

// `counters` layout (u64): [0..15] scratch (debug stamps) | one episodes-finished slot per 64-env group |
// STAT_GEN_SLOTS x {maps generated, whole-map retries}; mg_get_counters sums them on the host
constexpr int STAT_EPISODES = 16;
constexpr uint32_t STAT_GEN_SLOTS = 4096;

#ifdef MG_DEBUG_TIMING
// tuning aid (never built into the product library): cycle stamps of the first wave of block 0 -> counters[4..]
#define MG_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) A.counters[4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define MG_STAMP(k) do { } while (0)
#endif

MG_D uint64_t pick5(const uint64_t w[5], uint32_t k) { return k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : k == 3 ? w[3] : w[4]; }

constexpr int GEN_SBASE_BYTES = (int)GEN_SBASE_ENTRIES * 16;
constexpr int GEN_SCRATCH_BYTES = 64;   // generator state that must survive a restart from a checkpoint (MultiRoom's room lists), at the end
MG_HD int gen_wave_lds_bytes(int CS, int cap_words) { return CS + GEN_SBASE_BYTES + (cap_words + 4) * 4 + GEN_SCRATCH_BYTES; }

// wave-cooperative: all 64 lanes of one wave call this with the same `e`; `lds` = gen_wave_lds_bytes() of LDS
template <int GG, class RNG>
MG_D void generate_one(const GenArgs& A, RNG& rng, int e, uint32_t lane, uint8_t* lds) {
  const size_t N = (size_t)A.N;

  uint8_t* mygrid = lds;
  MG_STAMP(1);
  rng.load(A.rng, N, (size_t)e, lds + A.CS);
  MG_STAMP(2);
  if (A.rng_snap && lane < 5u) A.rng_snap[lane * N + (size_t)e] = pick5(rng.w_in, lane);
  GridRef g{ mygrid, A.gp.W, A.gp.H, (int)lane };
  for (int k = A.gp.W * A.gp.H + (int)lane; k < A.CS; k += 64) mygrid[k] = 0;
  GenResult out;
  // draw-budget loop: buffer `budget` draws, run the generator.  A pass that ran out of draws restarts from its
  // last checkpoint (GoToRedBall: the start of the current whole-map attempt) with a fresh buffer, or -- no
  // checkpoint passed -- is replayed from the start (same draws, same path) with twice the budget.  One refill
  // covers every DoorKey/Crossing episode; GoToRedBall (about 60 draws per attempt, 15.6 % of attempts rejected)
  // starts with three.
  const uint32_t cap = ((uint32_t)A.cap_words / RNG::kRefillWords) * RNG::kRefillWords;
  const uint32_t budget0 = A.gp.kind == 3 ? ((384u + RNG::kRefillWords - 1u) / RNG::kRefillWords) * RNG::kRefillWords : RNG::kRefillWords;
  uint32_t budget = budget0, retries_before = 0;
  out.resume = 0;
  for (;;) {
    budget = min(budget, cap);
    while (rng.limit < rng.off + budget) rng.refill();
    MG_STAMP(3);
    rng.begin_pass();
    // The generator parameters are made opaque per pass: otherwise every switch case's loop-invariant set-up is
    // hoisted out of this (rarely repeated) loop and all of it is live at once -- 160+ VGPRs instead of < 70, i.e.
    // 256 B/lane of scratch on every wave of a k_step launch under its register budget.
    GenParams gp = A.gp;
    gp.scratch_off = gen_wave_lds_bytes(A.CS, A.cap_words) - GEN_SCRATCH_BYTES;
    asm volatile("" : "+s"(gp.kind), "+s"(gp.W), "+s"(gp.H), "+s"(gp.start_x), "+s"(gp.start_y), "+s"(gp.start_dir));
    asm volatile("" : "+s"(gp.num_crossings), "+s"(gp.obstacle_cell), "+s"(gp.num_dists), "+s"(gp.strip2_row), "+s"(gp.room_size), "+s"(gp.random_length), "+s"(gp.scratch_off));
    g.W = gp.W; g.H = gp.H;
    asm volatile("" : "+v"(g.p), "+v"(g.lane));
    generate_episode<GG>(rng, g, gp, out);
    MG_STAMP(4);
    out.retries += retries_before;
    if (!rng.dead()) break;
    if (rng.ck != 0) { retries_before = out.retries; rng.rebase_to_checkpoint(); budget = budget0; out.resume = 1; continue; }
    if (budget >= cap) { out.failed = true; break; }
    budget *= 2u;
  }
  uint64_t w[5];
  rng.final_words(w);
  MG_STAMP(5);
  if (lane < 5u) A.rng[lane * N + (size_t)e] = pick5(w, lane);
  MG_WAVE_LDS_SYNC();
  uint4* dst = (uint4*)(A.dst_grid + (size_t)e * A.CS);
  for (int k = (int)lane; k < (A.CS >> 4); k += 64) dst[k] = ((const uint4*)mygrid)[k];
  if (lane == 0) {
    Agent ag; ag.x = out.ax; ag.y = out.ay; ag.dir = out.dir; ag.carry = 0; ag.step = 0; ag.mission = out.mission;
    ag.flags = (A.live && A.queue) ? FLAG_FRESH : 0u;
    A.dst_agent[e] = agent_pack(ag);
    if constexpr (GG != GG_LIGHT && GG != GG_ROOMS) { if (A.dst_aux) A.dst_aux[e] = out.aux; }     // no GG_LIGHT / GG_ROOMS level has an auxiliary word
    if (out.failed) atomicOr(A.err, (uint32_t)ERR_GENERATOR);
    unsigned long long* st = A.counters + A.stat_gen_off + 2u * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (STAT_GEN_SLOTS - 1u));
    atomicAdd(&st[0], 1ull);                                         // (mostly) private slot per generating wave
    if (out.retries) atomicAdd(&st[1], (unsigned long long)out.retries);
  }
  MG_STAMP(6);
  MG_WAVE_LDS_SYNC();
}

// stand-alone launch: 4 generating waves per workgroup (explicit resets, flushes of a pending refill queue)
constexpr int GEN_THREADS = 256;
template <class RNG>
__global__ void __launch_bounds__(GEN_THREADS) k_generate(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t lane = threadIdx.x & 63u;
  const int wave = (int)(threadIdx.x >> 6);
  MG_STAMP(0);
  if (A.zero_count && blockIdx.x == 0 && threadIdx.x == 0) *A.zero_count = 0u;
  RNG rng;
  rng.prefetch(lane);
  const int total = A.queue ? (int)uni32(*A.count) : A.N;
  uint8_t* lds = smem + wave * gen_wave_lds_bytes(A.CS, A.cap_words);
  const int nwaves = (int)gridDim.x * (GEN_THREADS / 64);
  for (int i = (int)blockIdx.x * (GEN_THREADS / 64) + wave; i < total; i += nwaves) {
    const int e = A.queue ? (int)uni32(A.queue[i]) : i;
    if (!A.queue && A.mask && !uni32(A.mask[e])) continue;
    generate_one<GG_ALL, RNG>(A, rng, e, lane, lds);
  }
}

// ======================================================================================================
// k_step: MiniGridEnv.step (minigrid_env.py:525-595) + RoomGridLevel.step/GoToInstr (roomgrid_level.py:87-104,
// verifier.py:309-316) + gen_obs (597-650: get_view_exts/slice/rotate_left/process_vis/encode) or
// FullyObsWrapper.observation (wrappers.py:419-426), with Gymnasium NEXT_STEP autoreset.
// MODE 0 = partial VxVx3 view, 1 = FullyObs WxHx3, 2 = one-hot partial view VxVx20, 3 = symbolic WxHx3,
// 4 = tile map for k_render (RGBImgPartialObsWrapper: VxV bytes; RGBImgObsWrapper: WxH bytes), byte = tile key * 2 + highlight.
// WPG = wavefronts per group of 64 envs (1, 2 or 4).  VT = 7 (default view, unrolled) or 15 (run-time V <= 15).
// GG = generator group compiled into the generator role (mg_gen.h; GG_NONE for levels whose reset draws nothing).
//
// One workgroup = 64 consecutive envs; lane l of EVERY wave is env l.  The kernel is VALU-issue bound (profiles/),
// so the split is chosen to minimise instructions while keeping the chip full: the per-env scalar dynamics (~150
// instructions) are recomputed by each wave; view rows / grid columns are dealt round-robin to the waves; the
// sequential process_vis pass runs in ONE wave and is shared through LDS; the host picks WPG so that a launch has
// >= ~4 waves per SIMD (small batches: 4, large batches: 1 = no redundant work at all).
// LDS: the 64 staged grids (coalesced 16 B/lane loads) between two guard bands so that out-of-grid view cells
// need no address clamp, per-env opacity rows, the visibility mask, the observation as final output bytes (copied
// out with 16 B/lane stores), and a 256-entry cell code -> (type,colour,state) table.
// ======================================================================================================
template <int MODE, int WPG, class RNG, int VT, int GG>
__global__ void __launch_bounds__(64 * WPG) __attribute__((amdgpu_waves_per_eu(7, 8)))   // <= 72 VGPRs: no scratch in any 7x7 variant; measured best
k_step(const StepParams P, const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int NT = 64 * WPG;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // ---- generator role: the FIRST gen_blocks workgroups refill the spare episodes that the PREVIOUS launch
  //      consumed (its refill queue), one generating wave per workgroup, concurrently with this launch's step
  //      groups.  Disjointness: an env consumed its spare in launch L-1 without stepping, so it cannot be due for
  //      a reset in launch L; the step groups of launch L therefore never read the spares written here, and launch
  //      L+1 starts after this one has completed. ----
  if (GG != GG_NONE && (int)blockIdx.x < P.gen_blocks) {
    MG_STAMP(0);
    __builtin_amdgcn_s_setprio(3);      // few, latency-critical scalar waves: issue ahead of the step waves
    if (blockIdx.x == 0 && lane == 0) *A.zero_count = 0u;
    // the queue slot, the queue length and the jump table are loaded together (one memory round trip, not three);
    // a slot beyond the queue's length holds a stale env id that is simply not used
    // Entry i of the queue goes to wave (i / gen_blocks) % WPG of workgroup i % gen_blocks: a short queue is served
    // by the wave 0s of as many workgroups as possible, a synchronized truncation burst by all waves.
    RNG rng;
    rng.prefetch((uint32_t)lane);
    const int first = (int)blockIdx.x + wave * P.gen_blocks;
    int e = first < A.N ? (int)A.queue[first] : 0;
    const int total = (int)uni32(*A.count);
    uint8_t* lds = smem + wave * gen_wave_lds_bytes(A.CS, A.cap_words);
    for (int i = first; i < total; i += P.gen_blocks * WPG) {
      if (i != first) e = (int)A.queue[i];
      generate_one<GG, RNG>(A, rng, (int)uni32((uint32_t)e), (uint32_t)lane, lds);
    }
    return;
  }
  const int env0 = ((int)blockIdx.x - P.gen_blocks) * 64;
  const int e = env0 + lane;
  const bool active = e < P.N;
  const int nvalid = min(64, P.N - env0);
  const int W = P.W, H = P.H, CS = P.CS, GS = P.GS;
  uint8_t* sgrid = smem + P.off_grid;
  uint8_t* strow = smem + P.off_trow;
  unsigned long long* svis = (unsigned long long*)(smem + P.off_vis);
  uint8_t* sT = smem + P.off_T;
  uint32_t* slut = (uint32_t*)(smem + P.off_lut);
  uint8_t* sact = smem + P.off_act;
  const bool reset_enabled = P.autoreset_next_step || P.phase == PHASE_OBSERVE;
  auto block_sync = [&]() {
    if constexpr (WPG == 1) { MG_WAVE_LDS_SYNC(); } else { __syncthreads(); }
  };

  // ---- every independent load is issued up front ----
  const uint64_t rec = active ? P.agent[e] : 0ull;
  // Every level-specific rule belongs to exactly one generator group (mg_create checks it), so a variant only carries
  // the rules its levels can have: GG_ROOMGRID GoTo / Unlock / Pickup, GG_LIGHT Fetch / GoToDoor / RedBlueDoors /
  // Memory, GG_NONE DynamicObstacles.
  uint64_t targets = 0;                         // BabyAI GoTo levels: tracked positions, issued with the other loads
  if constexpr (GG == GG_ROOMGRID) targets = ((P.rule == RULE_GOTO || P.rule == RULE_GOTOOBJ) && active) ? P.aux[e] : 0ull;
#pragma unroll
  for (int k = tid; k < 256; k += NT) slut[k] = MODE == 4 ? cell_tile_key((uint32_t)k) * 2u + 1u : cell_triple((uint32_t)k);
  if (wave == 0) sact[lane] = (uint8_t)((active && P.phase == PHASE_STEP) ? load_action(P, e) : (uint32_t)A_DONE);
  {
    // stage the 64 grids: 16 B per lane, fully coalesced; an env whose previous step ended its episode takes the
    // pre-generated spare episode instead (MiniGridEnv.reset, minigrid_env.py:119-157) and makes it the live grid
    const int cpe = CS >> 4;
    const int nchunks = nvalid * cpe;
    const uint4* live = (const uint4*)(P.grid + (size_t)env0 * CS);
    for (int c = tid; c < nchunks; c += NT) {
      const uint32_t el = ((uint32_t)c * P.cpe_magic) >> 20;
      const uint32_t part = (uint32_t)c - el * (uint32_t)cpe;
      const uint64_t srec = P.agent[env0 + el];
      uint4 v = live[c];
      if (((uint32_t)(srec >> 48) & FLAG_RESET_PENDING) && reset_enabled) {
        v = ((const uint4*)(P.spare_grid + (size_t)env0 * CS))[c];
        ((uint4*)(P.grid + (size_t)env0 * CS))[c] = v;
      }
      uint32_t* dst = (uint32_t*)(sgrid + el * GS + part * 16);
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
  }
  block_sync();

  Agent a = agent_unpack(rec);
  const uint8_t* mygrid = sgrid + lane * GS;
  uint32_t act = sact[lane];
  if constexpr (GG == GG_LIGHT) if (P.rule == RULE_MEMORY && act == A_PICKUP) act = A_TOGGLE;    // MemoryEnv.step (memory.py:151-153)
  if constexpr (GG == GG_NONE) if (P.rule == RULE_DYNOBS && act >= 3u) act = A_LEFT;             // "Invalid action" (dynamicobstacles.py:137-139)
  double reward = 0.0;
  uint32_t term = 0, trunc = 0, errbits = 0;
  bool rec_dirty = false;
  // The staged LDS grid is READ-ONLY after the barrier: the waves recompute the dynamics redundantly, so a wave
  // that wrote the toggled/picked/dropped cell back into LDS would be seen by a slower wave as its *input*.
  // Instead the one cell an action can change is patched on the fly.  It can only change under pickup/drop/toggle,
  // which leave the pose alone, so it is always the cell straight ahead: view cell (3,5).
  int dirty_idx = -1;              // linear index of the modified cell, -1 = none
  uint32_t dirty_code = 0;

  if (active) {
    if ((a.flags & FLAG_RESET_PENDING) && reset_enabled) {
      a = agent_unpack(P.spare_agent[e]);
      a.carry = 0; a.step = 0; a.flags = 0;
      rec_dirty = true;
      if constexpr (GG == GG_ROOMGRID) if ((P.rule == RULE_GOTO || P.rule == RULE_GOTOOBJ) && wave == 0) P.aux[e] = P.spare_aux[e];
      if (wave == 0 && !P.static_gen) {
        const uint32_t slot = atomicAdd(P.refill_count, 1u);
        P.refill_queue[slot] = (uint32_t)e;
      }
    } else if (a.flags & FLAG_FRESH) {
      a.flags &= ~(FLAG_FRESH | FLAG_NOT_CLEAR);       // drawn by the generator launch just before this one: observe only
      rec_dirty = true;
    } else if (P.phase == PHASE_STEP) {
      // ---- MiniGridEnv.step ----
      rec_dirty = true;
      a.step = min(a.step + 1u, 0xFFFFu);
      const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
      const bool inb = (unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H;
      if (!inb) errbits |= ERR_OOB;                                  // reference asserts (core/grid.py:74-78)
      const uint32_t fidx = inb ? (uint32_t)(fy * W + fx) : 0u;
      const uint32_t F = inb ? (uint32_t)mygrid[fidx] : (uint32_t)CELL_WALL_GREY;
      uint32_t newF = F;
      const uint32_t ftype = cell_type(F);
      bool success = false;
      if (act == A_LEFT) a.dir = (a.dir + 3u) & 3u;
      else if (act == A_RIGHT) a.dir = (a.dir + 1u) & 3u;
      else if (act == A_FORWARD) {
        if (cell_walkable(F)) { a.x = (uint32_t)fx; a.y = (uint32_t)fy; }
        if (ftype == T_GOAL) { term = 1; success = true; }
        if (ftype == T_LAVA) term = 1;
      } else if (act == A_PICKUP) {
        if (cell_pickable(F) && a.carry == 0) { a.carry = F; newF = CELL_EMPTY; }
      } else if (act == A_DROP) {
        if (F == CELL_EMPTY && a.carry != 0) { newF = a.carry; a.carry = 0; }
      } else if (act == A_TOGGLE) {
        newF = cell_toggle(F, a.carry);
      } else if (act != A_DONE) {
        errbits |= ERR_BAD_ACTION;                                   // reference raises ValueError (584-585)
      }
      if (newF != F && inb) {
        dirty_idx = (int)fidx; dirty_code = newF;
        if (wave == 0) P.grid[(size_t)e * CS + fidx] = (uint8_t)newF;
      }
      trunc = a.step >= (uint32_t)P.max_steps;
      if constexpr (GG == GG_ROOMGRID) if (P.rule == RULE_GOTO) {
        // RoomGridLevel.step (roomgrid_level.py:87-104) + GoToInstr.verify_action (verifier.py:309-316): success iff
        // the post-action front cell is one of the TRACKED POSITIONS of the described objects.  They are positions,
        // not objects: refreshed only at reset and after a drop (update_objs_poss), so they go stale while a target is
        // carried -- which only matters when a finished episode keeps being stepped (autoreset disabled).
        if (act == A_DROP) {
          // desc: rule_div 0 = fixed cell code (rule_cell), 1 = red/blue ball by mission id, 2 = (colour, type) by mission id
          const uint32_t m18 = a.mission % 18u;
          const uint32_t desc = P.rule_div == 0 ? (uint32_t)P.rule_cell
                              : P.rule_div == 1 ? make_cell(T_BALL, a.mission ? (uint32_t)C_BLUE : (uint32_t)C_RED)
                                                : make_cell(T_KEY + m18 % 3u, color_from_sorted(m18 / 3u));
          targets = 0;
          for (int k = 0; k < P.cells; k++) {
            const uint32_t c = k == dirty_idx ? dirty_code : (uint32_t)mygrid[k];
            targets |= (uint64_t)(c == desc) << k;
          }
          if (wave == 0) P.aux[e] = targets;
        }
        const int gx = (int)a.x + dir_dx(a.dir), gy = (int)a.y + dir_dy(a.dir);
        if ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && ((targets >> (gy * W + gx)) & 1ull)) { term = 1; success = true; }
      }
      if constexpr (GG == GG_ROOMGRID) if (P.rule == RULE_GOTOOBJ) {
        // GoToObjectEnv.step (gotoobject.py:137-153): toggle ends the episode; done ends it, rewarded when the agent
        // stands next to target_pos (the one-bit board drawn at reset)
        if (act == A_TOGGLE) term = 1;
        if (act == A_DONE) {
          const int ax = (int)a.x, ay = (int)a.y;          // interior cell: the four neighbours are inside the grid
          const uint64_t ring = (1ull << (ay * W + ax - 1)) | (1ull << (ay * W + ax + 1)) | (1ull << ((ay - 1) * W + ax)) | (1ull << ((ay + 1) * W + ax));
          term = 1; success = (targets & ring) != 0;
        }
      }
      if constexpr (GG == GG_LIGHT) if (P.rule == RULE_FETCH && a.carry != 0) {
        // FetchEnv.step (fetch.py:162-175): carrying anything ends the episode; the target (type, colour) is encoded
        // in the mission id = syntax*12 + COLOR_NAMES index*2 + (key 0 | ball 1)
        const uint32_t m12 = a.mission % 12u;
        const uint32_t target = make_cell((m12 & 1u) ? (uint32_t)T_BALL : (uint32_t)T_KEY, color_from_sorted(m12 >> 1));
        term = 1; success = a.carry == target;
      }
      if constexpr (GG == GG_ROOMGRID) if (P.rule == RULE_UNLOCK && act == A_TOGGLE) {
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
      if constexpr (GG == GG_ROOMGRID) if (P.rule == RULE_PICKUP && act == A_PICKUP && a.carry != 0) {
        // UnlockPickupEnv.step (unlockpickup.py:99-107) & co.: `self.carrying == self.obj`; the target is the only
        // object of its (type, colour): type = rule_cell, colour from the mission id / rule_div
        const uint32_t target = make_cell((uint32_t)P.rule_cell, color_from_sorted(a.mission / (uint32_t)P.rule_div));
        if (a.carry == target) { term = 1; success = true; }
      }
      if constexpr (GG == GG_ROOMS) if (P.rule == RULE_PICKUPDESC && act == A_PICKUP && a.carry != 0) {
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
      if constexpr (GG == GG_ROOMS) if (P.rule == RULE_OPENFRONT && act == A_TOGGLE) {
        // OpenInstr.verify_action (verifier.py:270-287): the cell in front is the described door (the level's only one)
        // and it is open after the toggle
        if (inb && cell_type(newF) == T_DOOR) { term = 1; success = true; }
      }
      if constexpr (GG == GG_LIGHT) if (P.rule == RULE_REDBLUE) {
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
      if constexpr (GG == GG_LIGHT) if (P.rule == RULE_MEMORY) {
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
      if constexpr (GG == GG_LIGHT) if (P.rule == RULE_GOTODOOR) {
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
      if (success) reward = a.step <= (uint32_t)P.max_steps ? P.reward_lut[a.step] : reward_exact(a.step, P.max_steps);
      if constexpr (GG == GG_NONE) if (P.rule == RULE_DYNOBS) {
        // DynamicObstaclesEnv.step (dynamicobstacles.py:162-165): walking into what WAS an obstacle or wall before the
        // obstacles moved (k_move_obstacles recorded it) costs -1 and ends the episode, whatever happened since
        if (act == A_FORWARD && (a.flags & FLAG_NOT_CLEAR)) { reward = -1.0; term = 1; }
        a.flags &= ~FLAG_NOT_CLEAR;
      }
      if (P.no_death_mask && term) {
        // NoDeath.step (wrappers.py:860-882): walking into (or ending the episode while standing in) a no-death
        // cell does not terminate; death_cost is added to the reward instead
        const bool going = act == A_FORWARD && F != CELL_EMPTY && ((P.no_death_mask >> cell_ref_type(F)) & 1);
        const uint32_t U = mygrid[(int)a.y * W + (int)a.x];
        const bool in_death = U != CELL_EMPTY && ((P.no_death_mask >> cell_ref_type(U)) & 1);
        if (going || in_death) { term = 0; reward = __dadd_rn(reward, P.death_cost); }
      }
      if ((term | trunc) && P.autoreset_next_step) {
        a.flags |= FLAG_RESET_PENDING;
        if (P.live_gen && wave == 0) {                 // drawn in place by the generator launch before the next step
          const uint32_t slot = atomicAdd(P.refill_count, 1u);
          P.refill_queue[slot] = (uint32_t)e;
        }
      }
    }
  }
  if (wave == 0 && P.phase == PHASE_STEP) {
    const unsigned long long fin = __ballot(active && (term | trunc));   // episodes finished in this group
    // statistics go to a slot owned by this workgroup: atomics contended on ONE line cost 4-12 us per launch here
    if (fin && lane == 0) atomicAdd(&P.counters[STAT_EPISODES + (env0 >> 6)], (unsigned long long)__popcll(fin));
  }

  // per-env scalar outputs, spread over the waves (each wave holds identical values)
  if (active) {
    if (wave == 0) { if (rec_dirty) P.agent[e] = agent_pack(a); if (errbits) atomicOr(P.err, errbits); }
    if (wave == 1 % WPG) P.reward[e] = reward;
    if (wave == 2 % WPG) { P.term[e] = (uint8_t)term; P.trunc[e] = (uint8_t)trunc; }
    if (wave == 3 % WPG) { P.dir_out[e] = (uint8_t)a.dir; P.mission_out[e] = (uint8_t)a.mission; }
  }

  const int obe = P.OBE;
  if (MODE == 0 || MODE == 2 || MODE == 4) {
    // ---- gen_obs_grid(V): closed form of get_view_exts + slice + rotate_left^(dir+1) (453-484, grid.py:110-143):
    //      view cell (vx,vy) is world cell agent + f*(V-1-vy) + r*(vx-V/2), f = DIR_TO_VEC[dir], r = (-f.y, f.x);
    //      outside the grid -> grey wall (grid.py:136-139).  wx depends on only one of vx/vy and wy on the other,
    //      so in-bounds-ness is (column mask)[vx] & (row mask)[vy].
    //      VT == 7: the reference's default view, fully unrolled.  VT == 15: ViewSizeWrapper (any odd V <= 15),
    //      same code with the loops guarded by the run-time V. ----
    const int V = VT == 7 ? 7 : P.view;
    const int HV = V >> 1;
    constexpr int RPW = (VT + WPG - 1) / WPG;                   // view rows per wave: rows wave, wave+WPG, ...
    const int fxv = dir_dx(a.dir), fyv = dir_dy(a.dir);
    const int rx = -fyv, ry = fxv;
    const bool horiz = fyv == 0;                                // facing +-x: wx moves with vy, wy with vx
    const uint32_t colmask = horiz ? inb_mask_v((int)a.y - HV * ry, ry, H, V) : inb_mask_v((int)a.x - HV * rx, rx, W, V);
    const uint32_t rowmask = horiz ? inb_mask_v((int)a.x + (V - 1) * fxv, -fxv, W, V) : inb_mask_v((int)a.y + (V - 1) * fyv, -fyv, H, V);
    const int SR = ry * W + rx;                                 // linear index step per vx
    const int SU = -(fyv * W + fxv);                            // linear index step per vy
    // may point outside this env's grid (into a neighbour's or a guard band): such cells are masked below
    const uint8_t* vbase = mygrid + ((int)a.y + (V - 1) * fyv - HV * ry) * W + ((int)a.x + (V - 1) * fxv - HV * rx);
    const uint32_t full = (1u << V) - 1u;
    uint32_t mycell[RPW][VT];
#pragma unroll
    for (int r = 0; r < RPW; r++) {
      const int vy = wave + WPG * r;
      if (vy < V) {
        const uint8_t* rowp = vbase + vy * SU;
        const uint32_t cm = ((rowmask >> vy) & 1u) ? colmask : 0u;
        uint32_t opq = 0;
#pragma unroll
        for (int vx = 0; vx < VT; vx++) {
          if (VT == 7 || vx < V) {
            const uint32_t raw = rowp[vx * SR];
            const uint32_t valid = 0u - ((cm >> vx) & 1u);
            uint32_t c = ((raw ^ CELL_WALL_GREY) & valid) ^ CELL_WALL_GREY;
            if (vx == HV && vy == V - 2) c = dirty_idx >= 0 ? dirty_code : c;   // the cell straight ahead
            mycell[r][vx] = c;
            opq |= (c >> 7) << vx;
          }
        }
        if (!P.see_through) {                                    // transparency bits of this view row
          if (VT == 7) strow[lane * 8 + vy] = (uint8_t)(~opq & 0x7Fu);
          else ((uint16_t*)strow)[lane * 16 + vy] = (uint16_t)(~opq & full);
        }
      }
    }
    // ---- process_vis (grid.py:291-328), bit-parallel rows bottom-up, in ONE wave; shared through LDS ----
    unsigned long long vis = ~0ull;                              // VT == 7: 49 bits, row j at bits 7j..7j+6
    if (!P.see_through) {
      block_sync();
      if (VT == 7) {
        if (wave == WPG - 1) {
          const uint2 tw = *(const uint2*)(strow + lane * 8);
          uint32_t m = 1u << (VIEW / 2);
          vis = 0;
#pragma unroll
          for (int j = VIEW - 1; j >= 0; j--) {
            const uint32_t t = ((j < 4 ? tw.x : tw.y) >> (8 * (j & 3))) & 0x7Fu;
            uint32_t vr, up;
            vis_row(m, t, &vr, &up);
            vis |= (unsigned long long)vr << (7 * j);
            m = up;
          }
          if (WPG > 1) svis[lane] = vis;
        }
        if (WPG > 1) { block_sync(); vis = svis[lane]; }
      } else {
        // wider views: one 16-bit mask per row, written back over the transparency rows
        if (wave == WPG - 1) {
          uint16_t* rows = (uint16_t*)strow + lane * 16;
          uint32_t m = 1u << HV;
          for (int j = V - 1; j >= 0; j--) {
            uint32_t vr, up;
            vis_row_n(m, rows[j], V, &vr, &up);
            rows[j] = (uint16_t)vr;
            m = up;
          }
        }
        block_sync();
      }
    }
    // ---- Grid.encode(vis_mask) (grid.py:244-268) straight into the output byte image [vx][vy][...]; invisible ->
    //      (0,0,0); the agent's own cell shows what it carries (minigrid_env.py:623-630).
    //      MODE 2: OneHotPartialObsWrapper (wrappers.py:267-284): 20 bytes per cell, one 1 in each of the type /
    //      colour / state groups. ----
    uint8_t* myT = sT + lane * obe;
#pragma unroll
    for (int r = 0; r < RPW; r++) {
      const int vy = wave + WPG * r;
      if (vy < V) {
        uint32_t vrow;
        if (VT == 7) vrow = (uint32_t)(vis >> (7 * vy)) & 0x7Fu;
        else vrow = P.see_through ? full : (uint32_t)((const uint16_t*)strow)[lane * 16 + vy];
#pragma unroll
        for (int vx = 0; vx < VT; vx++) {
          if (VT == 7 || vx < V) {
            uint32_t c = mycell[r][vx];
            if (vx == HV && vy == V - 1) c = a.carry ? a.carry : (uint32_t)CELL_EMPTY;
            if (MODE == 4) {
              // get_pov_render (minigrid_env.py:652-666): process_vis has blanked the invisible cells (grid.py:324-327),
              // so they are empty un-highlighted tiles (byte 0); visible ones are highlighted.  Image order [vy][vx].
              if (!P.rgb_full) myT[vy * V + vx] = (uint8_t)(slut[c] & (0u - ((vrow >> vx) & 1u)));
              continue;
            }
            const uint32_t tri = slut[c & (0u - ((vrow >> vx) & 1u))];
            if (MODE == 0) {
              uint8_t* o = myT + (vx * V + vy) * 3;
              o[0] = (uint8_t)tri; o[1] = (uint8_t)(tri >> 8); o[2] = (uint8_t)(tri >> 16);
            } else if (active) {
              // lanes past the batch end hold stale LDS "cells": their codes could index past the 20 bytes
              uint8_t* o = myT + (vx * V + vy) * 20;
              uint32_t* o4 = (uint32_t*)o;
              o4[0] = 0; o4[1] = 0; o4[2] = 0; o4[3] = 0; o4[4] = 0;
              o[tri & 0xFF] = 1; o[11 + ((tri >> 8) & 0xFF)] = 1; o[17 + (tri >> 16)] = 1;
            }
          }
        }
      }
    }
    if (MODE == 4 && VT == 7) {
      if (P.rgb_full) {
        // get_full_render (minigrid_env.py:668-714): every grid cell, highlighted where the agent's view sees it.
        // World cell (x, y) is view cell (HV + d.r, V-1 - d.f) with d = (x, y) - agent: the inverse of the gather above.
        const uint32_t hl_on = P.rgb_highlight ? 1u : 0u;
        for (int y = wave; y < H; y += WPG) {
          const int dy = y - (int)a.y;
#pragma unroll 4
          for (int x = 0; x < W; x++) {
            const int idx = y * W + x, dx = x - (int)a.x;
            uint32_t c = mygrid[idx];
            if (idx == dirty_idx) c = dirty_code;
            const int fwd = dx * fxv + dy * fyv, side = dx * rx + dy * ry + HV;
            const bool inside = (unsigned)fwd < (unsigned)V && (unsigned)side < (unsigned)V;
            const uint32_t bit = inside ? (uint32_t)(vis >> (7 * (V - 1 - fwd) + side)) & hl_on : 0u;
            myT[idx] = (uint8_t)(slut[c] - 1u + bit);
          }
        }
      }
    }
  } else {
    // ---- MODE 1: FullyObsWrapper.observation: grid.encode() in image[x][y] order, agent cell = (10, 0, dir)
    //      MODE 3: SymbolicObsWrapper.observation: (x, y, type or -1), agent cell type = 10 ----
    uint8_t* myT = sT + lane * obe;
    const int aidx = (int)a.y * W + (int)a.x;
    auto cell_tri = [&](int x, int y) -> uint32_t {
      const int idx = y * W + x;
      uint32_t c = mygrid[idx];
      if (idx == dirty_idx) c = dirty_code;
      if (MODE == 1) {
        if (idx == aidx) c = T_AGENT_MARK | (a.dir << 4);
        return slut[c];
      }
      const uint32_t t = idx == aidx ? (uint32_t)T_AGENT : (c == CELL_EMPTY ? 0xFFu : cell_ref_type(c));
      return (uint32_t)x | ((uint32_t)y << 8) | (t << 16);
    };
    auto put = [&](uint8_t* o, uint32_t tri) { o[0] = (uint8_t)tri; o[1] = (uint8_t)(tri >> 8); o[2] = (uint8_t)(tri >> 16); };
    for (int x = wave; x < W; x += WPG) {
      uint8_t* col = myT + x * H * 3;                           // column x of image[x][y][3]
      int y = 0;
      for (; y + 4 <= H; y += 4) {                              // four independent grid -> table -> store chains in flight
        const uint32_t t0 = cell_tri(x, y), t1 = cell_tri(x, y + 1), t2 = cell_tri(x, y + 2), t3 = cell_tri(x, y + 3);
        put(col + 3 * y, t0); put(col + 3 * y + 3, t1); put(col + 3 * y + 6, t2); put(col + 3 * y + 9, t3);
      }
      for (; y < H; y++) put(col + 3 * y, cell_tri(x, y));
    }
  }
  block_sync();

  // ---- the group's observations are one contiguous byte stream in LDS and in HBM: 16 B per lane per store ----
  {
    uint8_t* obase = P.obs + (size_t)env0 * (size_t)obe;          // 64*obe is a multiple of 16
    const int nbytes = nvalid * obe;
    const int nvec = nbytes >> 4;
    for (int c = tid; c < nvec; c += NT) ((uint4*)obase)[c] = ((const uint4*)sT)[c];
    for (int b = (nvec << 4) + tid; b < nbytes; b += NT) obase[b] = sT[b];   // ragged last group only
  }
}

// DynamicObstaclesEnv.step, the part before MiniGridEnv.step (dynamicobstacles.py:141-157): remember whether the
// front cell is occupied, then move every obstacle, in list order, to a random free cell of its 3x3 neighbourhood
// (place_obj with max_tries=100 on the ENV's stream; an obstacle that finds no place stays).  One lane per env,
// straight on the HBM state: this level's step consumes the stream, so it is kept out of k_step's register budget.
template <class RNG>
__global__ void k_move_obstacles(uint8_t* grid, uint64_t* agent, uint64_t* rng, uint64_t* obst, int N, int W, int H, int CS,
                                 int n_obst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  Agent a = agent_unpack(agent[e]);
  if (a.flags & (FLAG_RESET_PENDING | FLAG_FRESH)) return;          // no step for this env in the coming launch
  uint8_t* g = grid + (size_t)e * CS;
  const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
  const uint32_t F = ((unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H) ? (uint32_t)g[fy * W + fx] : (uint32_t)CELL_WALL_GREY;
  const bool not_clear = F != CELL_EMPTY && cell_type(F) != T_GOAL;
  RNG r;
  r.load(rng, (size_t)N, (size_t)e);
  uint64_t o = obst[e];
  for (int i = 0; i < n_obst; i++) {
    const int idx = (int)((o >> (8 * i)) & 0xFF), ox = idx % W, oy = idx / W;
    const int topx = max(ox - 1, 0), topy = max(oy - 1, 0), hx = min(topx + 3, W), hy = min(topy + 3, H);
    int tries = 0, nx = -1, ny = -1;
    for (;;) {
      if (tries > 100) break;                                       // RecursionError, swallowed by `except Exception`
      tries++;
      const int x = rand_int(r, topx, hx), y = rand_int(r, topy, hy);
      if (g[y * W + x] != CELL_EMPTY) continue;
      if (x == (int)a.x && y == (int)a.y) continue;
      nx = x; ny = y;
      break;
    }
    if (nx >= 0) {
      g[ny * W + nx] = (uint8_t)CELL_BALL_BLUE;
      g[idx] = (uint8_t)CELL_EMPTY;
      o = (o & ~(0xFFull << (8 * i))) | ((uint64_t)(ny * W + nx) << (8 * i));
    }
  }
  r.store(rng, (size_t)N, (size_t)e);
  obst[e] = o;
  a.flags = (a.flags & ~FLAG_NOT_CLEAR) | (not_clear ? FLAG_NOT_CLEAR : 0u);
  agent[e] = agent_pack(a);
}

// gymnasium.Env.reset(seed=s): np_random = Generator(PCG64(SeedSequence(s)))  (minigrid_env.py:125)
template <class RNG>
__global__ void k_seed(uint64_t* rng, const uint64_t* seeds, const uint8_t* mask, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  RNG r;
  r.seed(seeds[e]);
  r.store(rng, (size_t)N, (size_t)e);
}

// mark envs for an explicit reset() that continues their stream (consumes the spare)
__global__ void k_mark_pending(uint64_t* agent, const uint8_t* mask, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  Agent a = agent_unpack(agent[e]);
  a.flags |= FLAG_RESET_PENDING;
  agent[e] = agent_pack(a);
}

// ======================================================================================================
// k_render: RGBImgObsWrapper / RGBImgPartialObsWrapper (wrappers.py:287-380) = Grid.render (grid.py:200-242): the
// frame is a mosaic of pre-rendered tiles (mg_tiles.h).  Input: k_step's tile map (one byte per cell = tile key * 2 +
// highlight) and, for the full render, the agent record; output: [N][Ht*ts][Wt*ts][3] bytes.
//
// HBM-write bound (9-12 KB written per env against ~60 B read), so the kernel is organised around the store stream:
//  * A workgroup's EPW consecutive frames are ONE contiguous byte range, dealt out as 16 B chunks, thread t taking
//    chunks t, t + T, t + 2T, ... with T a multiple of the chunks per "period" (R pixel rows, R the smallest count
//    whose dwords divide by 4).  A thread's position inside its period -- which tile columns and which dword of the
//    tile row its four dwords come from -- is therefore loop-invariant; per chunk only the period index is
//    decomposed into env / tile row / pixel row, incrementally and with 24-bit multiplies.
//  * Tiles are read from LDS: the 102 agent-free tiles are staged once per workgroup (which then loops over groups
//    of EPW envs), the one agent tile of each env (cell kind x direction x highlight) once per env.
// Measured (profiles/r1_final/render_*.txt): the kernel runs at the speed of its own bare store loop; the write order
// (contiguous per workgroup vs. all workgroups sweeping adjacent frames) made no difference on MI355X.
// ======================================================================================================
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int RENDER_MAX_THREADS = 1024;          // 256 per workgroup while the LDS footprint lets >= 4 workgroups share a CU, else 1024
constexpr int STATIC_TILES = 2 * TILE_KEYS;          // [key][highlight]

struct RenderParams {
  const uint8_t* tilemap; const uint64_t* agent;
  const uint32_t* atlas_static;    // [key][hl][ts][ts*3/4] dwords
  const uint32_t* atlas_agent;     // [key][dir][hl][...]
  uint4* out;
  int N, Wt, Ht, cells, ts, full, epw, ngroups;
  int tile_dw, tdw_row, rowdw, R, cpp, ppe, t_active, pp;      // see above; ppe = periods per env, pp = periods per sweep
  int log2R; uint32_t magic_ts, magic_tdw;                     // R = 1 << log2R; magic_x = ceil(2^16 / x)
  int off_map;                                                 // LDS: [atlas dwords | u16 tile offsets per cell]
};

__global__ void __launch_bounds__(RENDER_MAX_THREADS) k_render(const RenderParams R) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* satlas = (uint32_t*)smem;
  uint16_t* smap = (uint16_t*)(smem + R.off_map);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = (int)blockDim.x;
  for (int i = tid; i < STATIC_TILES * R.tile_dw; i += nthreads) satlas[i] = R.atlas_static[i];

  // loop-invariant position of this thread's four dwords inside a period
  const bool worker = tid < R.t_active;
  const int cidx = tid % R.cpp, p0 = tid / R.cpp;
  int txj[4], srcj[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int dw = cidx * 4 + j, dr = dw / R.rowdw, col = dw - dr * R.rowdw;
    txj[j] = col / R.tdw_row;
    srcj[j] = dr * R.tdw_row + (col - txj[j] * R.tdw_row);
  }
  const int img_chunks = R.ppe * R.cpp;

  for (int g = blockIdx.x; g < R.ngroups; g += gridDim.x) {
    const int env0 = g * R.epw, nv = min(R.epw, R.N - env0);
    __syncthreads();                                                 // the previous group's blit is done with smap / the agent tiles
    for (int el = wave; el < nv; el += nthreads >> 6) {
      const int env = env0 + el;
      // the agent's cell: POV = bottom centre facing up (minigrid_env.py:659-663); full = its position and direction
      int cell = (R.Ht - 1) * R.Wt + (R.Wt >> 1), dir = 3;
      if (R.full) {
        const Agent a = agent_unpack(R.agent[env]);
        cell = (int)a.y * R.Wt + (int)a.x; dir = (int)a.dir;
      }
      const uint8_t* tm = R.tilemap + (size_t)env * R.cells;
      uint16_t* m = smap + el * R.cells;
      for (int k = lane; k < R.cells; k += 64) m[k] = (uint16_t)__umul24((uint32_t)tm[k], (uint32_t)R.tile_dw);
      MG_WAVE_LDS_SYNC();
      const uint32_t tb = (__umul24((uint32_t)m[cell], R.magic_tdw) >> 16);          // the tile byte under the agent
      const uint32_t* src = R.atlas_agent + (size_t)(((tb >> 1) * 4u + (uint32_t)dir) * 2u + (tb & 1u)) * R.tile_dw;
      uint32_t* dst = satlas + (STATIC_TILES + el) * R.tile_dw;
      for (int k = lane; k < R.tile_dw; k += 64) dst[k] = src[k];
      MG_WAVE_LDS_SYNC();
      if (lane == 0) m[cell] = (uint16_t)((STATIC_TILES + el) * R.tile_dw);
    }
    __syncthreads();
    if (worker) {
      // period p = p0, p0 + pp, ...: (env, period inside the env) advance by constant steps with one conditional
      // wrap; the rest is 24-bit multiplies of small numbers (full rate), no division
      u32x4* out = (u32x4*)R.out + (size_t)env0 * img_chunks + (uint32_t)(p0 * R.cpp + cidx);
      const uint32_t ostep = (uint32_t)(R.pp * R.cpp);
      const int total = nv * R.ppe, d_el = R.pp / R.ppe, d_pr = R.pp - d_el * R.ppe;
      int el = p0 / R.ppe, pr = p0 - el * R.ppe;
      int mb = el * R.cells;
      const int d_mb = d_el * R.cells;
#pragma unroll 2
      for (int p = p0; p < total; p += R.pp) {
        const uint32_t row0 = (uint32_t)pr << R.log2R;
        const uint32_t ty = __umul24(row0, R.magic_ts) >> 16;
        const uint32_t rowoff = __umul24(row0 - __umul24(ty, (uint32_t)R.ts), (uint32_t)R.tdw_row);
        const uint16_t* m = smap + mb + __umul24(ty, (uint32_t)R.Wt);
        u32x4 v;
        v.x = satlas[(uint32_t)m[txj[0]] + rowoff + srcj[0]];
        v.y = satlas[(uint32_t)m[txj[1]] + rowoff + srcj[1]];
        v.z = satlas[(uint32_t)m[txj[2]] + rowoff + srcj[2]];
        v.w = satlas[(uint32_t)m[txj[3]] + rowoff + srcj[3]];
        *out = v;
        out += ostep;
        pr += d_pr; mb += d_mb;
        if (pr >= R.ppe) { pr -= R.ppe; mb += R.cells; }
      }
    }
  }
}

}  // namespace mg
